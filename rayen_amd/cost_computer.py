"""``CostComputer``: the losses the reference's training harness evaluates on the layer's output.

Host-side mirror of ``examples/cost_computer.py`` (class ``CostComputer``, :21-138) so that a training script
written against the reference keeps working next to :class:`rayen_amd.constraint_module.ConstraintModule`: same
constructor (``CostComputer(cs)``), same method names and return values, buffers that follow ``.to(device)``.
It is plain PyTorch on whatever device ``y`` lives on -- a handful of batched products per training step, not
part of the projection hot path.  Differences in *how*: the inequality values are evaluated
family by family with batched products instead of a growing ``torch.cat`` (the reference's O(Q^2 B) copies,
SURVEY.md section 8a), and the LMI soft cost -- ``NotImplementedError`` in the reference (:105-106) -- stays that way.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import utils


def _quad(y, P, q, r):
    """``0.5 y'Py + q'y + r`` for ``y [B,k,1]`` and one (``[k,k]``) or a stack (``[Q,k,k]``) of forms -> ``[B,Q]``."""
    P = P.reshape(-1, P.shape[-2], P.shape[-1]).to(y)
    q = q.reshape(P.shape[0], -1).to(y)
    r = r.reshape(P.shape[0]).to(y)
    yv = y[:, :, 0]
    Py = torch.einsum("qij,bj->bqi", P, yv)
    return 0.5 * torch.einsum("bqi,bi->bq", Py, yv) + yv @ q.T + r


class CostComputer(nn.Module):
    def __init__(self, cs):
        super().__init__()
        if cs.has_quadratic_constraints:
            all_P, all_q, all_r = utils.getAllPqrFromQcs(cs.qcs)
            self.register_buffer("all_P", torch.Tensor(np.array(all_P)))
            self.register_buffer("all_q", torch.Tensor(np.array(all_q)))
            self.register_buffer("all_r", torch.Tensor(np.array(all_r)))
        if cs.has_soc_constraints:
            all_M, all_s, all_c, all_d = utils.getAllMscdFromSocs(cs.socs)
            self.register_buffer("all_M", torch.Tensor(np.array(all_M)))
            self.register_buffer("all_s", torch.Tensor(np.array(all_s)))
            self.register_buffer("all_c", torch.Tensor(np.array(all_c)))
            self.register_buffer("all_d", torch.Tensor(np.array(all_d)))
        self.register_buffer("A_p", torch.Tensor(cs.A_p))
        self.register_buffer("b_p", torch.Tensor(cs.b_p))
        self.register_buffer("yp", torch.Tensor(cs.yp))
        self.register_buffer("NA_E", torch.Tensor(cs.NA_E))
        self.register_buffer("z0", torch.Tensor(cs.z0))
        self.has_linear_ineq_constraints = cs.has_linear_ineq_constraints
        self.has_linear_eq_constraints = cs.has_linear_eq_constraints
        self.has_quadratic_constraints = cs.has_quadratic_constraints
        self.has_soc_constraints = cs.has_soc_constraints
        self.has_lmi_constraints = cs.has_lmi_constraints
        if self.has_linear_ineq_constraints:
            self.register_buffer("A1", torch.Tensor(cs.lc.A1))
            self.register_buffer("b1", torch.Tensor(cs.lc.b1))
        if self.has_linear_eq_constraints:
            self.register_buffer("A2", torch.Tensor(cs.lc.A2))
            self.register_buffer("b2", torch.Tensor(cs.lc.b2))

    def getyFromz(self, z):
        return self.NA_E @ z + self.yp

    def getInequalityValues(self, y):
        """``[B, m1 + Q + S]``: ``A1 y - b1``, every quadratic ``g_i(y)``, every cone ``||M y + s|| - c'y - d``
        (the stack cost_computer.py:69-103 builds; positive = violated)."""
        if self.has_lmi_constraints:
            raise NotImplementedError
        parts = []
        yv = y[:, :, 0]
        if self.has_linear_ineq_constraints:
            parts.append(yv @ self.A1.T - self.b1[:, 0])
        if self.has_quadratic_constraints:
            parts.append(_quad(y, self.all_P, self.all_q, self.all_r))
        if self.has_soc_constraints:
            My = torch.einsum("sij,bj->bsi", self.all_M, yv) + self.all_s[:, :, 0]
            parts.append(torch.linalg.vector_norm(My, dim=2) - yv @ self.all_c[:, :, 0].T - self.all_d[:, 0, 0])
        if not parts:
            return yv.new_zeros((y.shape[0], 0))
        return torch.cat(parts, dim=1)

    def getSumSoftCostAllSamples(self, y):
        soft_cost = torch.sum(torch.square(torch.relu(self.getInequalityValues(y))))
        if self.has_linear_eq_constraints:
            soft_cost = soft_cost + torch.sum(torch.square(y[:, :, 0] @ self.A2.T - self.b2[:, 0]))
        return soft_cost

    def getSumObjCostAllSamples(self, y, Pobj, qobj, robj):
        """``sum_b 0.5 y_b'P y_b + q'y_b + r`` with ONE objective (``P [k,k]``, ``q [k,1]``, ``r [1,1]``) or one per
        sample, as the reference's ``DataLoader`` hands them over (``P [B,k,k]``, ``q [B,k,1]``, ``r [B,1,1]``;
        examples/main.py:132-155, ``utils.quadExpression`` rayen/utils.py:228-242)."""
        if Pobj.ndim == 3 or qobj.ndim == 3 or robj.ndim == 3:
            B = y.shape[0]
            yv = y[:, :, 0]
            P = Pobj.to(y).expand(B, -1, -1) if Pobj.ndim == 3 else Pobj.to(y).unsqueeze(0).expand(B, -1, -1)
            q = qobj.to(y).reshape(-1, yv.shape[1]).expand(B, -1)
            r = robj.to(y).reshape(-1).expand(B)
            tmp = 0.5 * torch.einsum("bi,bij,bj->b", yv, P, yv) + torch.sum(q * yv, dim=1) + r
            if tmp.shape != (B,):
                raise RuntimeError(f"objective batch {tuple(Pobj.shape)} does not match {B} samples")
            return torch.sum(tmp)
        tmp = _quad(y, Pobj, qobj, robj)
        if tmp.shape != (y.shape[0], 1):
            raise RuntimeError(f"expected one objective, got P {tuple(Pobj.shape)}")
        return torch.sum(tmp)

    def getSumSupervisedCostAllSamples(self, y, y_predicted):
        return torch.sum(torch.square(y - y_predicted))

    def getSumLossAllSamples(self, params, y, y_predicted, Pobj, qobj, robj, isTesting=False):
        loss = 0.0
        if params['use_supervised']:
            loss = loss + self.getSumSupervisedCostAllSamples(y, y_predicted)
        else:
            loss = loss + self.getSumObjCostAllSamples(y_predicted, Pobj, qobj, robj)
        if (not isTesting) and params['weight_soft_cost'] > 0:
            loss = loss + params['weight_soft_cost'] * self.getSumSoftCostAllSamples(y_predicted)
        return loss
