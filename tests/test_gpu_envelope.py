"""The shapes of the reference's own timing sweep (examples/scripts/time_analysis.py:57-192: up to 10^4 linear or
quadratic rows, LMIs to 300 x 300) on a HIP device: which route serves each, that every one of them RUNS and meets the
oracle, and that a route which is not a hand-written kernel says so (RuntimeWarning) instead of passing silently.

Routes: the matrix-core / lane kernels (n up to several hundred, any number of rows), the wave-per-sample LMI kernels
(r <= ~190 fp32), and -- beyond what the kernels stage -- the packed form evaluated with device library calls
(rocBLAS GEMM ``v W'`` + rocSOLVER ``eigvalsh`` through torch: rayen_amd/eager.py), forward and backward."""
import warnings

import numpy as np
import pytest
import torch

from helpers import csd_from_cs, rel_err_rows
from oracle import rayen_oracle as oracle
from rayen_amd import _lib, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _layer64(raw):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        cs = workloads.build_constraints(raw)
        return cs, ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)


def _run(layer, x):
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        y = layer(x)
    return y, [w for w in caught if issubclass(w.category, RuntimeWarning)]


def test_ten_thousand_linear_rows_on_the_kernels():
    """time_analysis.py:62-63 goes to 10^4 linear constraints: a kernel serves them (no detour, no warning)."""
    raw = workloads.random_lin_quad_soc(k=32, m=10000, n_quad=0, n_soc=0, seed=5)
    cs, layer = _layer64(raw)
    x = torch.empty(512, cs.n, 1, dtype=torch.float64).uniform_(-1, 1)
    y, warned = _run(layer, x.cuda())
    assert not warned and not layer._hip_unsupported
    want = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x)
    assert np.max(rel_err_rows(y.cpu().numpy()[:, :, 0], want.numpy()[:, :, 0])) <= 1e-9
    layer32 = ConstraintModule(cs, create_map=False).cuda()
    y32, warned = _run(layer32, x.float().cuda())
    assert not warned
    assert np.max(rel_err_rows(y32.cpu().numpy()[:, :, 0], want.numpy()[:, :, 0])) <= 1e-5


@pytest.mark.eager_detour
@pytest.mark.parametrize("r", [250, 300])
def test_lmi_beyond_one_waves_lds_runs_on_the_device_libraries(r):
    """time_analysis.py:159-160 ends at 300 x 300: no kernel holds that matrix; the module says so once and evaluates
    the packed form with the device's libraries -- same answers as the reference's op sequence."""
    raw = workloads.random_lmi(6, r, seed=r)
    cs, layer = _layer64(raw)
    x = torch.empty(24, cs.n, 1, dtype=torch.float64).uniform_(-1, 1)
    y, warned = _run(layer, x.cuda())
    assert y.is_cuda and len(warned) == 1 and layer._hip_unsupported
    want = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x)
    assert np.max(rel_err_rows(y.cpu().numpy()[:, :, 0], want.numpy()[:, :, 0])) <= 1e-9
    assert cs.getMaxViolation(y.cpu().numpy()[:, :, 0]) <= 1e-9


@pytest.mark.eager_detour
def test_backward_of_a_wide_set_on_the_wide_route_and_on_the_loud_detour(monkeypatch):
    """n = 400 with quadratics and cones.  Default: forward and backward on the wide route (vendor GEMMs + the products
    epilogue / coefficient kernels), silently.  With RAYEN_WIDE_ROUTE=0 the forward is the lane kernel's, whose backward
    does not stage n = 400: the gradient then comes from autograd through the packed evaluator -- with a warning.
    Both equal autograd through the reference's op sequence."""
    raw = workloads.random_lin_quad_soc(k=400, m=60, n_quad=2, n_soc=1, r_M=30, seed=9)
    cs, layer = _layer64(raw)
    x = torch.empty(48, cs.n, 1, dtype=torch.float64).uniform_(-1, 1)
    w = torch.empty(48, cs.k, 1, dtype=torch.float64).uniform_(-1, 1)
    x2 = x.clone().requires_grad_(True)
    (oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x2) * w).sum().backward()

    def grad_and_warnings():
        xg = x.cuda().requires_grad_(True)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            (layer(xg) * w.cuda()).sum().backward()
        err = (xg.grad.cpu() - x2.grad).abs().amax(dim=(1, 2)) / x2.grad.abs().amax(dim=(1, 2)).clamp_min(1e-30)
        return float(err.max()), [c for c in caught if issubclass(c.category, RuntimeWarning)]

    # (kinks -- ties of the arg-max, kappa = 1 -- are measure-zero for these random directions at fp64)
    err, warned = grad_and_warnings()
    assert err <= 1e-7 and not warned and _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PRODUCTS
    monkeypatch.setenv("RAYEN_WIDE_ROUTE", "0")
    err, warned = grad_and_warnings()
    assert err <= 1e-7 and len(warned) == 1 and _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LANE


# --------------------------------------------------------------------------- the wide route: library GEMM + products epilogue
def _wide_sets():
    rng = np.random.default_rng(3)
    a = workloads.random_lin_quad_soc(k=200, m=700, n_quad=3, n_soc=2, r_M=150, seed=31)           # dense forms, NA_E = I
    b = workloads.random_lin_quad_soc(k=300, m=120, n_quad=2, n_soc=1, r_M=40, seed=32)            # + equalities: n = 260
    b["A2"] = rng.uniform(-1, 1, size=(40, 300))
    b["b2"] = np.zeros((40, 1))
    c = workloads.corridor_like(k=260, n_eq=20, m=400, n_quad=30, rank=4, seed=33)                 # low-rank factors, equalities
    d = workloads.random_lin_quad_soc(k=1000, m=3000, n_quad=0, n_soc=0, seed=34)                  # time_analysis.py:62-63
    return {"dense_n200": a, "eq_n260": b, "lowrank_n240": c, "lin_k1000": d}


@pytest.mark.parametrize("name", ["dense_n200", "eq_n260", "lowrank_n240", "lin_k1000"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-9)])
def test_wide_route_against_oracle_and_the_lane_kernel(name, dtype, tol):
    """n beyond the matrix-core kernels: T = v W_ext' on the vendor GEMM, then ONE kernel over T
    (rayen_ray_project_from_products_*).  Against the reference's op sequence, and against the lane-per-sample kernel
    on the same pack (kappa, the active record, y)."""
    from rayen_amd import ops
    raw = _wide_sets()[name]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    gen = torch.Generator().manual_seed(4)
    B = 1500
    x = torch.empty(B, cs.n, 1, dtype=torch.float32).uniform_(-1, 1, generator=gen).to(dtype)
    x[:4] *= 1e-4
    x[4] = 0
    y, warned = _run(layer, x.cuda())
    assert not warned and _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PRODUCTS
    want = oracle.forward(oracle.precompute(csd_from_cs(cs), dtype), x).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y.cpu().numpy()[:, :, 0], want)) <= tol
    floor = 1e-6 if dtype == torch.float32 else 1e-10
    assert cs.getMaxViolation(y.cpu().numpy()[:, :, 0]) <= max(floor, 3 * cs.getMaxViolation(want))
    assert np.allclose(y[4, :, 0].cpu().numpy(), cs.y0[:, 0], atol=1e-6)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    v = x[:, :, 0].cuda()
    yw, kw, aw = ops.project_raw(v, dp)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PRODUCTS
    yl, kl, al = ops.project_raw(v, dp, force_generic=True)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_LANE
    ktol = 2e-5 if dtype == torch.float32 else 1e-9
    assert torch.allclose(kw, kl, rtol=ktol, atol=ktol)
    same = (aw == al).all(dim=1)
    assert float(same.float().mean()) > 0.995          # (ties of the arg-max may fall either way)
    assert np.max(rel_err_rows(yw.cpu().numpy(), yl.cpu().numpy())) <= 2 * tol
    # backward on the wide route (coefficient kernel + one GEMM) against autograd through the oracle at fp64 away from
    # kinks, and against the lane backward where that kernel stages the set
    from helpers import kink_mask
    g = torch.empty(B, cs.k, dtype=dtype).uniform_(-1, 1, generator=gen)
    gw = ops.backward_raw(v, kw, aw, g.cuda(), dp)
    xo = x.double().clone().requires_grad_(True)
    (oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), xo)[:, :, 0] * g.double()).sum().backward()
    keep = ~kink_mask(oracle, cs, x.double(), 1e-6 if dtype == torch.float64 else 1e-3)
    keep[:5] = False
    gerr = (gw.cpu().double() - xo.grad[:, :, 0]).abs().amax(dim=1) / xo.grad[:, :, 0].abs().amax(dim=1).clamp_min(1e-30)
    assert keep.sum() > B // 2
    assert float(gerr[torch.as_tensor(keep)].max()) <= (2e-4 if dtype == torch.float32 else 1e-8)
    try:
        gl = ops.backward_raw(v, kl, al, g.cuda(), dp, force_generic=True)
        lerr = (gw - gl).abs().amax(dim=1) / gl.abs().amax(dim=1).clamp_min(1e-30)
        assert float(lerr[same & torch.as_tensor(keep, device="cuda")].max()) <= (4e-4 if dtype == torch.float32 else 1e-8)
    except _lib.RayenError as err:
        assert err.code == _lib.E_UNSUPPORTED          # (the lane backward does not stage this n)
    # wider input rows (only the first n columns are read), a strided output, kappa alone, and the empty batch
    wide_in = torch.cat((v, torch.full((B, 3), 7.0, dtype=dtype, device="cuda")), dim=1)
    buf = torch.zeros(B, cs.k + 5, dtype=dtype, device="cuda")
    y2, _, _ = ops.project_raw(wide_in, dp, want_active=False, want_kappa=False, out=buf)
    assert torch.equal(buf[:, :cs.k], yw) and float(buf[:, cs.k:].abs().max()) == 0.0
    _, k3, _ = ops.project_raw(v, dp, want_y=False)
    assert torch.equal(k3, kw)
    y0, k0, a0 = ops.project_raw(v[:0], dp)
    assert y0.shape == (0, cs.k) and k0.shape == (0,)
    # non-finite rows stay non-finite, raise the flag, and touch nobody else
    vb = v.clone()
    vb[7, 3] = float("nan")
    vb[9, 0] = float("inf")
    dp.nan_flag.zero_()
    yb, _, _ = ops.project_raw(vb, dp)
    assert int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    assert not torch.isfinite(yb[7]).all() and not torch.isfinite(yb[9]).all()
    keep = torch.ones(B, dtype=torch.bool, device="cuda")
    keep[[7, 9]] = False
    assert torch.equal(yb[keep], yw[keep])
