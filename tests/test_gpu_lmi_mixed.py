"""Quadratics / cones NEXT TO an LMI beyond what the lane kernels hold (~30 x 30): until round 5 these sets left the C ABI for
the device's libraries.  Now two launches (rayen_abi.hip::mixed_forward): the lane-per-sample kernel evaluates everything but
the LMI, the workgroup-per-sample kernel (rayen_lmi_block.h) the LMI on top of it -- forward and backward.  This suite runs
with RAYEN_STRICT_HIP=1 (conftest): a detour would raise.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from helpers import csd_from_cs, rel_err_rows
from oracle import rayen_oracle as oracle
from rayen_amd import _lib, ops, workloads
from test_gpu_lmi_wave import _check_backward, _layer

pytestmark = pytest.mark.gpu


def _mixed(k, r, m, n_quad, n_soc, scale, seed, n_eq=0):
    """random_lin_quad_soc + an r x r LMI whose generators are scaled so that every family clips some samples"""
    rng = np.random.default_rng(seed)
    raw = workloads.random_lin_quad_soc(k, m, n_quad, n_soc, r_M=min(k, 6), seed=seed)
    F = []
    for _ in range(k):
        T = rng.uniform(-1, 1, size=(r, r))
        F.append(scale * (T + T.T) / 2)
    T = rng.uniform(-1, 1, size=(r, r))
    F.append(T @ T.T + 0.5 * np.eye(r))
    raw["F"] = F
    if n_eq:
        raw["A2"] = rng.uniform(-1, 1, size=(n_eq, k))
        raw["b2"] = np.zeros((n_eq, 1))
    return raw


CASES = {
    "r60_all": dict(k=10, r=60, m=12, n_quad=2, n_soc=2, scale=0.7, seed=1),
    "r100_quad_soc": dict(k=8, r=100, m=0, n_quad=1, n_soc=1, scale=0.6, seed=2),
    "r40_lin_quad_eq": dict(k=12, r=40, m=20, n_quad=3, n_soc=0, scale=1.5, seed=3, n_eq=2),
    "r150": dict(k=6, r=150, m=5, n_quad=1, n_soc=1, scale=0.6, seed=4),
    "r290_head_columns": dict(k=5, r=290, m=4, n_quad=1, n_soc=1, scale=0.35, seed=5),     # fp32 only (r <= 212 in fp64)
}


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_lmi_next_to_quadratics_and_cones(name, dtype):
    if dtype == torch.float64 and CASES[name]["r"] > 212:
        pytest.skip("fp64: the workgroup kernel holds r <= 212")
    raw = _mixed(**CASES[name])
    r = CASES[name]["r"]
    cs, layer = _layer(raw, dtype)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(8)
    B = 64 if r <= 100 else 28
    x = torch.empty(B, cs.n).uniform_(-2.0, 2.0, generator=gen)
    x[:2] *= 1e-4                                             # interior
    x[2] = 0.0
    xd = x.to(dtype).cuda()
    y, kappa, active = ops.project_raw(xd, dp, want_active=True)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LMI_BLOCK
    buf64 = oracle.precompute(csd_from_cs(cs), torch.float64)
    xr = x.double().unsqueeze(2).requires_grad_(True)
    y_true_t = oracle.forward(buf64, xr)
    y_true = y_true_t.detach().numpy()[:, :, 0]
    err = rel_err_rows(y.cpu().double().numpy(), y_true)
    if dtype == torch.float64:
        assert err.max() <= 1e-9, (name, err.max())
    else:
        y32 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x.unsqueeze(2)).numpy()[:, :, 0]
        theirs = rel_err_rows(y32.astype(np.float64), y_true).max()
        assert err.max() <= max(1e-5, 2.0 * theirs), (name, err.max(), theirs)
    assert cs.getMaxViolation(y.cpu().double().numpy()) <= (1e-9 if dtype == torch.float64 else 2e-4)
    assert np.allclose(y[2].cpu().double().numpy(), cs.y0[:, 0], atol=1e-12 if dtype == torch.float64 else 1e-6)
    # every family sets kappa somewhere: the LMI (the last segment) and something else, and both clip
    seg = active[:, 0].cpu().numpy()
    clipped = kappa.cpu().numpy() > 1.0
    lmi_seg = int(seg.max())
    assert (seg[clipped] == lmi_seg).sum() >= 3 and ((seg[clipped] != lmi_seg) & (seg[clipped] >= 0)).sum() >= 3, np.bincount(seg[seg >= 0])
    # the module's path (no kappa wanted: the maximum over the other families travels in column 0 of y) gives the same rows
    y_mod = layer(xd.unsqueeze(2))
    assert not layer._hip_unsupported and torch.equal(y_mod[:, :, 0], y)
    y_nok, _, _ = ops.project_raw(xd, dp, want_active=False, want_kappa=False)
    assert torch.equal(y_nok, y)
    for b in (1, 7):
        assert torch.equal(ops.project_raw(xd[:b].contiguous(), dp)[0], y[:b])
    # ---- backward: lane kernel for every sample, then the workgroup kernel on the samples the LMI clipped
    _check_backward(cs, buf64, x, xd, xr, y_true_t, kappa, active, dp, dtype, r, gen, name)
    xg = xd.unsqueeze(2).clone().requires_grad_(True)
    layer(xg).sum().backward()
    assert not layer._hip_unsupported and torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_the_old_head_on_such_a_set(dtype):
    """RAYEN_old (CM:460-466) on a set with quadratics and cones next to a large LMI: kappa is the same, so the lane kernel
    runs without the head (it writes no y on this route) and the workgroup kernel takes the step 1 / (||v|| e^beta + kappa);
    backward: the lane kernel's old-head gradient for every sample, the workgroup kernel's for the samples whose kappa is
    the LMI's.  Forward against the oracle's RAYEN_old, backward against autograd through it."""
    from rayen_amd.constraint_module import ConstraintModule
    raw = _mixed(**CASES["r60_all"])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, method="RAYEN_old", create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    gen = torch.Generator().manual_seed(3)
    B = 64
    x = torch.empty(B, cs.n + 1, 1).uniform_(-1.5, 1.5, generator=gen)
    xg = x.to(dtype).cuda().requires_grad_(True)
    y = layer(xg)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LMI_BLOCK and not layer._hip_unsupported
    xr = x.double().requires_grad_(True)
    y_true_t = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), xr, method="RAYEN_old")
    err = rel_err_rows(y.detach().cpu().double().numpy()[:, :, 0], y_true_t.detach().numpy()[:, :, 0])
    assert err.max() <= (1e-9 if dtype == torch.float64 else 2e-5), err.max()
    assert cs.getMaxViolation(y.detach().cpu().double().numpy()[:, :, 0]) <= (1e-9 if dtype == torch.float64 else 2e-4)
    w = torch.empty(B, cs.k, 1).uniform_(-1, 1, generator=gen)
    (y * w.to(dtype).cuda()).sum().backward()
    (y_true_t * w.double()).sum().backward()
    got, want = xg.grad.cpu().double().numpy()[:, :, 0], xr.grad.numpy()[:, :, 0]
    gerr = np.abs(got - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-30)
    tol = 1e-7 if dtype == torch.float64 else 3e-3
    assert np.all(np.isfinite(got)) and (gerr > tol).sum() <= max(2, 0.05 * B), np.sort(gerr)[-5:]
    assert np.median(gerr) <= tol * 0.1


@pytest.mark.parametrize("name", ["r60_all", "r40_lin_quad_eq"])
def test_compute_kappa_and_a_full_grid_of_workgroups_on_such_a_set(name):
    """(round 6, the advisor's two findings on round 5's kernels.)  (i) ``computeKappa`` -- the reference's public helper
    CM:351 -- on a set the workgroup kernels serve: they always write y, so the module hands them a scratch y instead of
    raising.  (ii) The module's default inference call (no kappa wanted) parks the other families' kappa in column 0 of y:
    with MANY MORE samples than workgroups every wave must have taken its copy before thread 0 stores y[b][0] -- the rows must
    equal, bit for bit, those of the call with a separate kappa buffer, in every repetition."""
    raw = _mixed(**CASES[name])
    cs, layer = _layer(raw, torch.float32)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(21)
    B = 6000
    xd = torch.empty(B, cs.n).uniform_(-2.0, 2.0, generator=gen).cuda()
    y, kappa, _ = ops.project_raw(xd, dp, want_active=True)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LMI_BLOCK
    for _ in range(5):
        y_nok, _, _ = ops.project_raw(xd, dp, want_active=False, want_kappa=False)
        assert torch.equal(y_nok, y)
    got = layer.computeKappa(xd[:512].unsqueeze(2))
    assert not layer._hip_unsupported and got.shape == (512, 1, 1)
    assert torch.equal(got[:, 0, 0], kappa[:512])
    buf64 = oracle.precompute(csd_from_cs(cs), torch.float64)
    want = oracle.compute_kappa(buf64, xd[:128].cpu().double().unsqueeze(2))[:, 0, 0].numpy()
    assert np.max(np.abs(got[:128, 0, 0].cpu().double().numpy() - want) / np.maximum(np.abs(want), 1e-3)) <= 2e-5


def test_compute_kappa_on_a_large_lmi_alone():
    """``computeKappa`` on [linear rows] + one LMI of the workgroup kernels, fused route and products route (32 generators on)."""
    for k, r in ((10, 64), (40, 48)):
        raw = workloads.random_lmi(k, r, seed=5)
        cs, layer = _layer(raw, torch.float32)
        gen = torch.Generator().manual_seed(2)
        x = torch.empty(300, cs.n, 1).uniform_(-2.0, 2.0, generator=gen)
        got = layer.computeKappa(x.cuda())
        assert not layer._hip_unsupported
        want = oracle.compute_kappa(oracle.precompute(csd_from_cs(cs), torch.float64), x.double())[:, 0, 0].numpy()
        err = np.abs(got[:, 0, 0].cpu().double().numpy() - want) / np.maximum(np.abs(want), 1e-3)
        assert err.max() <= 2e-5, (k, r, err.max())

