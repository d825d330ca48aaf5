"""The packed projection in plain torch ops -- the product's own evaluator for tensors that do not live on an MI355X.

``rayen_amd`` computes the projection of rayen/constraint_module.py:351-474 from ONE row matrix ``W`` and a
segment table (``rayen_amd.pack.pack_constants``).  The HIP kernels behind ``include/rayen_hip.h`` consume that
form on a gfx950 device; this module evaluates the SAME form with ``torch`` tensor operations, on whatever device
and dtype the caller's tensors have, and is differentiable through autograd.  It exists for what the reference does
on the host (the reference's ``examples/test_layer.py:70-117`` runs the layer on CPU tensors) and for shapes no
kernel serves; it is never taken for a tensor on a HIP device that a kernel serves, and it is NOT the test oracle
(``oracle/`` restates the reference op for op from its own buffers; nothing here imports it).

=========  ===========================================  ======================================
segment    rows of ``T = v W'``                           candidate of kappa
=========  ===========================================  ======================================
LIN        ``D v``                                        ``max_i T_i``                     CM:353
QUAD_SYM   ``phi.v`` ; ``G v``                            ``phi.v + sqrt(max(v'Gv, 0))``    CM:374
QUAD_FAC   ``phi.v`` ; ``U v``                            ``phi.v + ||U v||``
SOC        ``c.v``, ``beta'M v`` ; ``M v``                larger root of ``a'x^2+b'x+c'``   CM:383-399
LMI        packed lower triangle of ``sum_a v_a G_a``     ``lambda_max``                    CM:401-449
=========  ===========================================  ======================================

``kappa = relu(max over segments)`` and ``y = y0 + NA_E v / max(1, kappa)`` (CM:468-474 by homogeneity: every
candidate is of degree 1 in ``v``); the ``RAYEN_old`` head takes the step ``1/(exp(beta) + kappa(v/|v|))``
(CM:460-466).  Documented deviations are the kernels' own: a cone the ray never meets contributes 0.
"""
from __future__ import annotations

import torch

from . import _lib

SEG_LIN, SEG_QUAD_SYM, SEG_QUAD_FAC, SEG_SOC, SEG_LMI = (_lib.SEG_LIN, _lib.SEG_QUAD_SYM, _lib.SEG_QUAD_FAC,
                                                         _lib.SEG_SOC, _lib.SEG_LMI)


class PackedEvaluator:
    """Constants of one layer (``PackedConstants``) as tensors of one dtype on one device."""

    def __init__(self, consts, dtype, device):
        self.k, self.n = consts.k, consts.n
        self.dtype, self.device = dtype, torch.device(device)
        as_t = lambda a: torch.as_tensor(a, dtype=torch.float64).to(device=self.device, dtype=dtype)   # noqa: E731
        self.Wt = as_t(consts.W).t().contiguous()             # [n, rows]
        self.NA_Et = as_t(consts.NA_E).t().contiguous()       # [n, k]
        self.y0 = as_t(consts.y0).reshape(1, self.k)
        self.identity = bool(consts.out_identity)
        self.segments = list(consts.segments)
        self._tril = {}
        for s in self.segments:
            if s.type == SEG_LMI and s.dim not in self._tril:
                self._tril[s.dim] = torch.tril_indices(s.dim, s.dim, device=self.device)

    # -------------------------------------------------------------------------------------------- kappa
    def candidates(self, v):
        """``v [B, n]`` -> ``[B, n_segments]``: every segment's candidate of kappa (before the relu)."""
        T = v @ self.Wt
        cols = []
        for s in self.segments:
            main = T[:, s.row0:s.row0 + s.nrows]
            if s.type == SEG_LIN:
                val = main.max(dim=1).values
            elif s.type == SEG_QUAD_SYM:
                val = T[:, s.aux_row] + torch.sqrt(torch.clamp((main * v).sum(dim=1), min=0.0))
            elif s.type == SEG_QUAD_FAC:
                val = T[:, s.aux_row] + torch.linalg.vector_norm(main, dim=1)
            elif s.type == SEG_SOC:
                cr, br = T[:, s.aux_row], T[:, s.aux_row + 1]
                cp = (main * main).sum(dim=1) - cr * cr
                bp = 2.0 * br - 2.0 * s.f0 * cr
                disc = bp * bp - 4.0 * s.f1 * cp
                root = torch.sqrt(torch.clamp(disc, min=0.0))
                val = torch.maximum((-bp - root) / (2.0 * s.f1), (-bp + root) / (2.0 * s.f1))
                val = torch.where(disc >= 0, val, torch.zeros_like(val))
            elif s.type == SEG_LMI:
                r = s.dim
                il, jl = self._tril[r]
                A = v.new_zeros((v.shape[0], r, r))
                A[:, il, jl] = main
                A = A + torch.transpose(torch.tril(A, -1), 1, 2)
                val = torch.linalg.eigvalsh(A)[:, -1] if v.shape[0] else v.new_zeros((0,))
            else:  # pragma: no cover
                raise ValueError(f"unknown segment type {s.type}")
            cols.append(val)
        if not cols:
            return v.new_zeros((v.shape[0], 0))
        return torch.stack(cols, dim=1)

    def kappa(self, v, want_active=False):
        cand = self.candidates(v)
        if cand.shape[1] == 0:
            kap = v.new_zeros((v.shape[0],))
            return (kap, torch.full((v.shape[0],), -1, dtype=torch.int64, device=v.device)) if want_active else kap
        top, idx = cand.max(dim=1)
        kap = torch.clamp(top, min=0.0)
        if want_active:
            return kap, torch.where(top > 0, idx, torch.full_like(idx, -1))
        return kap

    # -------------------------------------------------------------------------------------------- the projection
    def project(self, v, old_head=False):
        """``v [B, >= n (+1)]`` -> ``(y [B, k], kappa [B])``."""
        n = self.n
        d = v[:, :n]
        if old_head:
            d_bar = d / torch.clamp(torch.linalg.vector_norm(d, dim=1, keepdim=True), min=1e-12)
            kap = self.kappa(d_bar)
            step = 1.0 / (torch.exp(v[:, n]) + kap)
            move = d_bar * step.unsqueeze(1)
        else:
            kap = self.kappa(d)
            move = d / torch.clamp(kap, min=1.0).unsqueeze(1)
        y = self.y0 + (move if self.identity else move @ self.NA_Et)
        return y, kap


def evaluator_for(module, v):
    """The module's evaluator for tensors like ``v`` (built once per (dtype, device))."""
    cache = module.__dict__.setdefault("_eager", {})
    compute = v.dtype if v.dtype in (torch.float32, torch.float64) else torch.float32
    key = (compute, v.device)
    ev = cache.get(key)
    if ev is None:
        ev = cache[key] = PackedEvaluator(module.packed_constants(), compute, v.device)
    return ev


def project(module, v, old_head=False):
    """``(y, kappa)`` at ``v``'s dtype and device; 16-bit inputs are computed in fp32."""
    ev = evaluator_for(module, v)
    y, kap = ev.project(v.to(ev.dtype), old_head=old_head)
    return y.to(v.dtype), kap.to(v.dtype)
