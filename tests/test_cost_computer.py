"""``rayen_amd.cost_computer.CostComputer`` (the harness-side losses, examples/cost_computer.py:21-138) against the
formulas written out sample by sample, and against the residual metric of the constraint classes."""
import numpy as np
import torch

from rayen_amd import workloads
from rayen_amd.cost_computer import CostComputer


def _one_by_one(raw, y):
    """sum over samples of relu(inequality)^2 (+ equality residual^2), one constraint and one sample at a time."""
    total = 0.0
    for row in y:
        vals = []
        if raw["A1"] is not None:
            vals += list(raw["A1"] @ row - raw["b1"][:, 0])
        for P, q, r in zip(raw["P"], raw["q"], raw["r"]):
            vals.append(0.5 * row @ P @ row + q[:, 0] @ row + r[0, 0])
        for M, s, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"]):
            vals.append(np.linalg.norm(M @ row + s[:, 0]) - c[:, 0] @ row - d[0, 0])
        total += sum(max(v, 0.0) ** 2 for v in vals)
        if raw["A2"] is not None:
            total += float(np.sum((raw["A2"] @ row - raw["b2"][:, 0]) ** 2))
    return total


def test_soft_cost_objective_and_loss():
    torch.set_default_dtype(torch.float64)
    try:
        for raw in (workloads.make_raw("c2", seed=1), workloads.corridor_like(k=12, n_eq=3, m=20, n_quad=3, rank=2, seed=2),
                    workloads.random_lin_quad_soc(k=7, m=5, n_quad=1, n_soc=2, seed=3)):
            cs = workloads.build_constraints(raw)
            cc = CostComputer(cs)
            rng = np.random.default_rng(0)
            y = rng.uniform(-2, 2, size=(40, cs.k))
            yt = torch.tensor(y).unsqueeze(2)
            assert abs(cc.getSumSoftCostAllSamples(yt).item() - _one_by_one(raw, y)) <= 1e-9 * max(1.0, _one_by_one(raw, y))
            # feasible points cost nothing: the interior point, repeated
            y0 = torch.tensor(np.repeat(cs.y0.T, 5, axis=0)).unsqueeze(2)
            assert cc.getSumSoftCostAllSamples(y0).item() <= 1e-18
            # the stacked inequality values agree with the residual metric of the constraint classes
            vals = cc.getInequalityValues(yt).numpy()
            res = cs.getResiduals(y)
            worst = np.max(np.stack([res[key] for key in ("lin_ineq", "quad", "soc") if key in res], axis=1), axis=1)
            assert np.allclose(vals.max(axis=1), worst, atol=1e-12)
            P = rng.uniform(-1, 1, size=(cs.k, cs.k)); P = P @ P.T
            q = rng.uniform(-1, 1, size=(cs.k, 1)); r = np.array([[0.3]])
            want = sum(0.5 * row @ P @ row + q[:, 0] @ row + 0.3 for row in y)
            got = cc.getSumObjCostAllSamples(yt, torch.tensor(P), torch.tensor(q), torch.tensor(r)).item()
            assert abs(got - want) <= 1e-9 * abs(want)
            params = {"use_supervised": False, "weight_soft_cost": 10.0}
            loss = cc.getSumLossAllSamples(params, yt, yt, torch.tensor(P), torch.tensor(q), torch.tensor(r))
            assert abs(loss.item() - (want + 10.0 * _one_by_one(raw, y))) <= 1e-8 * abs(loss.item())
            sup = cc.getSumLossAllSamples({"use_supervised": True, "weight_soft_cost": 0.0}, yt + 1.0, yt, None, None, None)
            assert abs(sup.item() - y.size) <= 1e-9
    finally:
        torch.set_default_dtype(torch.float32)


def test_objective_with_one_form_per_sample():
    """The reference's harness passes DataLoader-batched objectives (examples/main.py:132-155: P [B,k,k], q [B,k,1],
    r [B,1,1]); every sample is paired with ITS objective (rayen/utils.py:228-242), never with the others'."""
    torch.set_default_dtype(torch.float64)
    try:
        raw = workloads.random_lin_quad_soc(k=6, m=4, n_quad=1, n_soc=1, seed=5)
        cc = CostComputer(workloads.build_constraints(raw))
        rng = np.random.default_rng(1)
        B, k = 5, 6
        y = rng.uniform(-2, 2, size=(B, k))
        T = rng.uniform(-1, 1, size=(B, k, k))
        P = T @ np.transpose(T, (0, 2, 1))
        q = rng.uniform(-1, 1, size=(B, k, 1))
        r = rng.uniform(-1, 1, size=(B, 1, 1))
        want = sum(0.5 * y[b] @ P[b] @ y[b] + q[b, :, 0] @ y[b] + r[b, 0, 0] for b in range(B))
        yt = torch.tensor(y).unsqueeze(2)
        got = cc.getSumObjCostAllSamples(yt, torch.tensor(P), torch.tensor(q), torch.tensor(r)).item()
        assert abs(got - want) <= 1e-12 * max(1.0, abs(want))
        # a shared P with per-sample q, r (the reference's broadcasting accepts it too)
        want2 = sum(0.5 * y[b] @ P[0] @ y[b] + q[b, :, 0] @ y[b] + r[b, 0, 0] for b in range(B))
        got2 = cc.getSumObjCostAllSamples(yt, torch.tensor(P[0]), torch.tensor(q), torch.tensor(r)).item()
        assert abs(got2 - want2) <= 1e-12 * max(1.0, abs(want2))
        # gradients reach y
        yt.requires_grad_(True)
        cc.getSumObjCostAllSamples(yt, torch.tensor(P), torch.tensor(q), torch.tensor(r)).backward()
        g = np.stack([P[b] @ y[b] + q[b, :, 0] for b in range(B)])
        assert np.allclose(yt.grad[:, :, 0].numpy(), g, atol=1e-12)
    finally:
        torch.set_default_dtype(torch.float32)
