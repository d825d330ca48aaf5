"""LMIs beyond the sizes of the four-lanes-per-sample kernel (32 x 32 fp32 / 24 x 24 fp64) and of the lane-per-sample
kernels (~30 x 30): one wave per sample with the matrix in LDS (rayen_amd/csrc/rayen_lmi_wave.h) -- Householder
tridiagonalisation, Sturm-count multisection over the 64 lanes; the backward maps the tridiagonal eigenvector back
through the reflectors.  The reference (rayen/constraint_module.py:401-449) handles any size and its own timing sweep
(examples/scripts/time_analysis.py:159-160) starts at 100 x 100.  Forward against the oracle (fp64: 1e-9; fp32: the
north_star's 1e-5 or twice what LAPACK's fp32 eigvalsh -- the reference's arithmetic -- leaves against the fp64 truth),
backward against autograd through the fp64 oracle's eigvalsh.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from helpers import csd_from_cs, rel_err_rows
from oracle import rayen_oracle as oracle
from rayen_amd import _lib, ops, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _case(k, r, m, n_eq, seed):
    rng = np.random.default_rng(seed)
    raw = workloads.random_lmi(k, r, seed=seed)
    F = []
    for _ in range(k):
        T = rng.uniform(-1, 1, size=(r, r))
        F.append((T + T.T) / 2)
    T = rng.uniform(-1, 1, size=(r, r))
    F.append(T @ T.T + 0.5 * np.eye(r))
    raw["F"] = F
    if m:
        raw["A1"] = rng.uniform(-1, 1, size=(m, k))
        raw["b1"] = rng.uniform(0.1, 1.0, size=(m, 1))
    if n_eq:
        raw["A2"] = rng.uniform(-1, 1, size=(n_eq, k))
        raw["b2"] = np.zeros((n_eq, 1))                      # y0 = 0 satisfies them
    return raw


CASES = {
    "r30_eq": dict(k=7, r=30, m=0, n_eq=2, seed=1),          # fp64: beyond the quad kernel's 24 x 24
    "r33": dict(k=6, r=33, m=0, n_eq=0, seed=2),             # the first size no other fp32 kernel takes
    "r48_lin": dict(k=9, r=48, m=30, n_eq=0, seed=3),
    "r64_lin_eq": dict(k=12, r=64, m=70, n_eq=3, seed=4),    # 70 linear rows: two row chunks per wave
    "r100": dict(k=10, r=100, m=0, n_eq=0, seed=5),          # the reference's own sweep starts here
    "r128_k70": dict(k=70, r=128, m=5, n_eq=0, seed=6),      # n > 64 as well; two rows per lane
}


def _check_backward(cs, buf64, x, xd, xr, y_true_t, kappa, active, dp, dtype, r, gen, name):
    """grad_v of the kernels against autograd through the fp64 oracle (eigvalsh), kinks set aside."""
    B = x.shape[0]
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
    (y_true_t[:, :, 0] * g.double()).sum().backward()
    want = xr.grad[:, :, 0].numpy()
    got = ops.backward_raw(xd, kappa, active, g.to(dtype).cuda(), dp).cpu().double().numpy()
    assert np.all(np.isfinite(got))
    size = np.maximum(np.abs(want).max(axis=1), 1e-30)
    gerr = np.abs(got - want).max(axis=1) / size
    gerr[2] = 0.0                                             # v = 0: eigvalsh of the zero matrix has no derivative worth comparing
    # kinks: the two largest eigenvalues (nearly) tie, or kappa within rounding of 1 / of the linear rows' maximum
    terms = oracle.compute_kappa(buf64, torch.nn.functional.normalize(x.double().unsqueeze(2), dim=1), terms=True).numpy()
    lam_gap = np.abs(terms[:, -1] - terms[:, -2]) / np.maximum(np.abs(terms[:, -1]), 1e-30)
    kap = kappa.cpu().double().numpy()
    kink = (lam_gap < (1e-3 if dtype == torch.float32 else 1e-7)) | (np.abs(kap - 1.0) < 1e-4)
    if terms.shape[1] > 2:
        top_lin = terms[:, :-2].max(axis=1)
        kink |= np.abs(top_lin - np.maximum(terms[:, -1], 0.0)) < 1e-4 * np.maximum(np.abs(top_lin), 1e-30)
    tol = 2e-3 if dtype == torch.float32 else 1e-7
    # (nearly repeated top eigenvalues that are no kink yet: what a backward-stable eigen-solver delivers,
    # eps r ||S|| / gap, next to the flat tolerance)
    eps = 6e-8 if dtype == torch.float32 else 1.1e-16
    bound = np.maximum(tol, 8.0 * r * eps / np.maximum(lam_gap, 1e-30))
    bad = (~kink) & ~(gerr <= bound)
    assert not bad.any(), (name, int(bad.sum()), np.flatnonzero(bad)[:5], gerr[bad][:5], bound[bad][:5])
    assert kink.sum() <= max(3, 0.05 * B)
    return got


def _layer(raw, dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        return cs, ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("name", list(CASES))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("kernel", ["block", "wave"])
def test_large_lmi_forward_and_backward(name, dtype, kernel, monkeypatch):
    """``kernel``: the forward on the workgroup-per-sample kernel (rayen_lmi_block.h, round 5: the default wherever it
    serves) or, pinned with RAYEN_LMI_BLOCK=0, on the wave-per-sample kernel; the same switch moves the backward."""
    if kernel == "wave" and name not in ("r30_eq", "r33", "r48_lin"):
        pytest.skip("the wave kernel is the default only for fp64 matrices up to 44 x 44 since round 5: three sizes stand for it")
    monkeypatch.setenv("RAYEN_LMI_BLOCK", "1" if kernel == "block" else "0")
    raw = _case(**CASES[name])
    r = CASES[name]["r"]
    cs, layer = _layer(raw, dtype)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(6)
    B = 160 if r <= 48 else (96 if r <= 64 else 40)           # (the fp64 oracle's eigvalsh + autograd on the CPU sets the pace)
    x = torch.empty(B, cs.n).uniform_(-2.0, 2.0, generator=gen)
    x[:2] *= 1e-4                                             # interior
    x[2] = 0.0
    xd = x.to(dtype).cuda()
    y, kappa, active = ops.project_raw(xd, dp, want_active=True)
    fam = _lib.load().rayen_last_forward_kernel()
    if name == "r30_eq" and dtype == torch.float32:
        assert fam == _lib.KERNEL_LMI_QUAD                    # (the small kernels keep what they can hold)
    else:
        assert fam == (_lib.KERNEL_LMI_BLOCK if kernel == "block" else _lib.KERNEL_LMI_WAVE), (name, fam)
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
    # feasibility of what came out (fp64 residuals)
    assert cs.getMaxViolation(y.cpu().double().numpy()) <= (1e-9 if dtype == torch.float64 else 2e-4)
    assert np.allclose(y[2].cpu().double().numpy(), cs.y0[:, 0], atol=1e-12 if dtype == torch.float64 else 1e-6)

    _check_backward(cs, buf64, x, xd, xr, y_true_t, kappa, active, dp, dtype, r, gen, name)


BIG = {
    "r200_lin": dict(k=8, r=200, m=40, n_eq=0, seed=11),     # beyond the wave kernel's LDS in fp32 (r <= ~190) and at the
    "r196_eq": dict(k=9, r=196, m=0, n_eq=2, seed=12),       # block kernel's limit in fp64 with the whole triangle in LDS
    "r250": dict(k=6, r=250, m=0, n_eq=0, seed=13),
    "r280_lin": dict(k=5, r=280, m=10, n_eq=0, seed=14),     # the largest packed triangle a workgroup's LDS holds in fp32 (281)
    # beyond the LDS: the first 24 (fp32) / 16 (fp64) columns in registers (rayen_lmi_block.h, head_phase)
    "r282": dict(k=5, r=282, m=0, n_eq=0, seed=15),          # the first size that needs them
    "r300": dict(k=6, r=300, m=0, n_eq=0, seed=16),          # the end of the reference's sweep (time_analysis.py:157-160)
    "r303_lin": dict(k=4, r=303, m=10, n_eq=0, seed=17),     # the largest the backward holds in fp32
    "r210_eq": dict(k=9, r=210, m=0, n_eq=2, seed=18),       # fp64 with 16 columns in registers
    # register columns so that TWO 512-thread workgroups share a compute unit's LDS (fp32 198 .. 220, fp64 139 .. ~152)
    "r212_lin": dict(k=7, r=212, m=6, n_eq=0, seed=19),
    "r150": dict(k=6, r=150, m=0, n_eq=0, seed=20),
}


@pytest.mark.parametrize("name,dtype", [("r200_lin", torch.float32), ("r196_eq", torch.float64),
                                         ("r280_lin", torch.float32), ("r282", torch.float32),
                                         ("r300", torch.float32), ("r303_lin", torch.float32), ("r210_eq", torch.float64),
                                         ("r212_lin", torch.float32), ("r150", torch.float64)])
def test_matrices_only_the_block_kernel_holds(name, dtype):
    """Forward AND backward of LMIs up to 303 x 303 (fp32) / 210 x 210 (fp64) on hand-written kernels -- the reference's own
    sweep ends at 300 x 300 (time_analysis.py:157-160); rounds 3-4 sent everything beyond ~190 / ~135 to rocSOLVER through
    the packed torch evaluator.  This suite runs with RAYEN_STRICT_HIP=1 (conftest): a detour would raise, and the
    wave-per-sample kernels refuse these sizes (their full r x r storage does not fit)."""
    raw = _case(**BIG[name])
    r = BIG[name]["r"]
    cs, layer = _layer(raw, dtype)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(2)
    B = 28                                                    # (the fp64 oracle's eigvalsh + autograd on the CPU sets the pace)
    x = torch.empty(B, cs.n, 1).uniform_(-2.0, 2.0, generator=gen)
    x[:2] *= 1e-4
    x[2] = 0.0
    y = layer(x.to(dtype).cuda())
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LMI_BLOCK and not layer._hip_unsupported
    buf64 = oracle.precompute(csd_from_cs(cs), torch.float64)
    xr = x.double().clone().requires_grad_(True)
    y_true_t = oracle.forward(buf64, xr)
    y_true = y_true_t.detach().numpy()[:, :, 0]
    err = rel_err_rows(y.cpu().double().numpy()[:, :, 0], y_true)
    if dtype == torch.float64:
        assert err.max() <= 1e-9, (name, err.max())
    else:
        y32 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x).numpy()[:, :, 0]
        theirs = rel_err_rows(y32.astype(np.float64), y_true).max()
        assert err.max() <= max(1e-5, 2.0 * theirs), (name, err.max(), theirs)
    assert cs.getMaxViolation(y.cpu().double().numpy()[:, :, 0]) <= (1e-9 if dtype == torch.float64 else 2e-4)
    # the batch does not matter (persistent workgroups, one sample after the other), nor does a second launch
    y2 = layer(x[:7].to(dtype).cuda())
    assert torch.equal(y2, y[:7]) and torch.equal(layer(x.to(dtype).cuda()), y)
    # ---- backward: the kernel's grad_v against autograd through the fp64 oracle, and through the module's autograd
    xd = x[:, :, 0].to(dtype).cuda().contiguous()
    yk, kappa, active = ops.project_raw(xd, dp, want_active=True)
    assert torch.equal(yk, y[:, :, 0])
    got = _check_backward(cs, buf64, x[:, :, 0], xd, xr, y_true_t, kappa, active, dp, dtype, r, gen, name)
    assert torch.equal(ops.backward_raw(xd[:7].contiguous(), kappa[:7].contiguous(), active[:7].contiguous(),
                                        torch.ones(7, cs.k, dtype=dtype, device="cuda"), dp),
                       ops.backward_raw(xd, kappa, active, torch.ones(B, cs.k, dtype=dtype, device="cuda"), dp)[:7])
    xg = x.to(dtype).cuda().requires_grad_(True)
    layer(xg).sum().backward()
    assert not layer._hip_unsupported and torch.isfinite(xg.grad).all() and got.shape == (B, cs.n)


@pytest.mark.parametrize("kernel", ["block", "wave"])
def test_wave_kernel_small_and_ragged_batches_and_nan_rows(kernel, monkeypatch):
    monkeypatch.setenv("RAYEN_LMI_BLOCK", "1" if kernel == "block" else "0")
    raw = _case(**CASES["r48_lin"])
    cs, layer = _layer(raw, torch.float32)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(9)
    x = torch.empty(67, cs.n).uniform_(-2, 2, generator=gen).cuda()
    y_all, k_all, _ = ops.project_raw(x, dp)
    assert _lib.load().rayen_last_forward_kernel() == (_lib.KERNEL_LMI_BLOCK if kernel == "block" else _lib.KERNEL_LMI_WAVE)
    for B in (1, 5, 64):
        y, kap, _ = ops.project_raw(x[:B].contiguous(), dp)
        assert torch.equal(y, y_all[:B]) and torch.equal(kap, k_all[:B])
    dp.nan_flag.zero_()
    xb = x.clone()
    xb[7, 1] = float("nan")
    yb, _, _ = ops.project_raw(xb, dp)
    assert int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    keep = torch.ones(67, dtype=torch.bool, device="cuda")
    keep[7] = False
    assert torch.equal(yb[keep], y_all[keep])


MANY = {
    "r50_k160_lin_eq": dict(k=160, r=50, m=10, n_eq=2, seed=31),      # NA_E != I: its products ride behind the rows of W
    "r120_k200": dict(k=200, r=120, m=0, n_eq=0, seed=32),
    "r290_k140_lin": dict(k=140, r=290, m=5, n_eq=0, seed=33),        # register columns (fp32 only)
}


@pytest.mark.parametrize("name", list(MANY))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_many_generators_take_the_products_route(name, dtype, monkeypatch):
    """n >= 129 (fp64: 65): S(v) for the whole batch is ONE library GEMM, T = v W_ext' (ops._wide_route), and the
    workgroup-per-sample kernels read S(v), D v and NA_E v from its rows (rayen_ray_project_from_products_*); the backward
    leaves the row of coefficients for the second GEMM (rayen_ray_project_bwd_coefficients_*).  Same answers as the fused
    kernels (RAYEN_WIDE_ROUTE=0), which form S(v) sample by sample."""
    r = MANY[name]["r"]
    if dtype == torch.float64 and r > 212:
        pytest.skip("fp64: the workgroup kernel holds r <= 212")
    raw = _case(**MANY[name])
    cs, layer = _layer(raw, dtype)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    assert dp.products_matrix(dtype) is not None
    gen = torch.Generator().manual_seed(4)
    B = 24
    x = torch.empty(B, cs.n).uniform_(-2.0, 2.0, generator=gen)
    x[:2] *= 1e-4
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
    got = _check_backward(cs, buf64, x, xd, xr, y_true_t, kappa, active, dp, dtype, r, gen, name)
    # the fused kernels on the same inputs
    monkeypatch.setenv("RAYEN_WIDE_ROUTE", "0")
    y_f, kappa_f, active_f = ops.project_raw(xd, dp, want_active=True)
    assert torch.equal(active_f, active)
    tol = 1e-12 if dtype == torch.float64 else 2e-5
    assert float((y_f - y).abs().max()) <= tol * max(1.0, float(y.abs().max()))
    g = torch.ones(B, cs.k, dtype=dtype, device="cuda")
    gv_f = ops.backward_raw(xd, kappa_f, active_f, g, dp)
    monkeypatch.delenv("RAYEN_WIDE_ROUTE")
    gv_p = ops.backward_raw(xd, kappa, active, g, dp)
    scale = float(gv_f.abs().max())
    assert float((gv_f - gv_p).abs().max()) <= (1e-9 if dtype == torch.float64 else 5e-3) * scale and got.shape == (B, cs.n)
    # the module: forward + autograd, no detour
    xg = xd.unsqueeze(2).clone().requires_grad_(True)
    layer(xg).sum().backward()
    assert not layer._hip_unsupported and torch.isfinite(xg.grad).all()
    # a NaN row raises the flag and touches no other row; smaller batches give the same rows
    dp.nan_flag.zero_()
    xb = xd.clone()
    xb[5, 1] = float("nan")
    yb, _, _ = ops.project_raw(xb, dp)
    assert int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    keep = torch.ones(B, dtype=torch.bool, device="cuda")
    keep[5] = False
    assert torch.equal(yb[keep], y[keep])
    for b in (1, 7):       # (not bit for bit: the library picks its GEMM by the batch)
        part = ops.project_raw(xd[:b].contiguous(), dp)[0]
        assert float((part - y[:b]).abs().max()) <= tol * max(1.0, float(y.abs().max()))


OLD_HEAD = {
    "r64_lin": dict(k=9, r=64, m=12, n_eq=0, seed=41),
    "r100_eq": dict(k=10, r=100, m=0, n_eq=2, seed=42),
    "r290": dict(k=5, r=290, m=0, n_eq=0, seed=43),            # register columns (fp32 only)
}


@pytest.mark.parametrize("name", list(OLD_HEAD))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_the_old_head_on_the_workgroup_kernels(name, dtype):
    """method='RAYEN_old' (CM:460-466: y = y0 + N v_bar / (e^beta + kappa(v_bar)), beta = column n of the input) for LMIs
    beyond the four-lane kernel's: the same kappa, another step, and gradients for ||v|| and beta -- forward against the
    oracle's RAYEN_old, backward against autograd through it (strict mode: the device libraries would raise)."""
    r = OLD_HEAD[name]["r"]
    if dtype == torch.float64 and r > 212:
        pytest.skip("fp64: the workgroup kernel holds r <= 212")
    raw = _case(**OLD_HEAD[name])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, method="RAYEN_old", create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    gen = torch.Generator().manual_seed(12)
    B = 28
    x = torch.empty(B, cs.n + 1, 1).uniform_(-1.5, 1.5, generator=gen)
    xg = x.to(dtype).cuda().requires_grad_(True)
    y = layer(xg)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LMI_BLOCK and not layer._hip_unsupported
    buf64 = oracle.precompute(csd_from_cs(cs), torch.float64)
    xr = x.double().requires_grad_(True)
    y_true_t = oracle.forward(buf64, xr, method="RAYEN_old")
    err = rel_err_rows(y.detach().cpu().double().numpy()[:, :, 0], y_true_t.detach().numpy()[:, :, 0])
    assert err.max() <= (1e-9 if dtype == torch.float64 else 2e-5), (name, err.max())
    assert cs.getMaxViolation(y.detach().cpu().double().numpy()[:, :, 0]) <= (1e-9 if dtype == torch.float64 else 2e-4)
    w = torch.empty(B, cs.k, 1).uniform_(-1, 1, generator=gen)
    (y * w.to(dtype).cuda()).sum().backward()
    (y_true_t * w.double()).sum().backward()
    got, want = xg.grad.cpu().double().numpy()[:, :, 0], xr.grad.numpy()[:, :, 0]
    assert np.all(np.isfinite(got))
    gerr = np.abs(got - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-30)
    tol = 1e-7 if dtype == torch.float64 else 3e-3
    # (kinks -- the two largest eigenvalues nearly tied, a linear row level with the LMI -- are a few rows at most)
    assert (gerr > tol).sum() <= max(2, 0.05 * B), (name, np.sort(gerr)[-5:])
    assert np.median(gerr) <= tol * 0.1
