"""Backward of the fused op (HIP kernel) against autograd through the CPU oracle (fp64)."""
import numpy as np
import pytest
import torch

from helpers import lmi_gradient_bound, csd_from_cs, load_golden, kink_mask
from oracle import rayen_oracle as oracle
from rayen_amd import workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _cases():
    mixed_raw, _, _ = load_golden("config_mixed")          # lin + eq + quad + SOC + LMI, n < k
    return {
        "c1": workloads.make_raw("c1"),
        "c2": workloads.make_raw("c2", seed=31),
        "c3": workloads.make_raw("c3", seed=32),
        "c4": workloads.make_raw("c4", seed=33),
        "c5r": workloads.corridor_like(k=20, n_eq=6, m=24, n_quad=5, rank=2, seed=34),
        "mixed": mixed_raw,
    }


# Gradient tolerances.  Away from the kinks of kappa (helpers.kink_mask: ties of the arg-max, kappa = 1, kappa = 0,
# a nearly repeated top LMI eigenvalue -- identified on the fp64 oracle, not budgeted) EVERY sample is held to
# GRAD_TOL of the fp64 autograd gradient, per-sample inf-norm relative; in fp32 a sample may exceed it only where
# the reference's own fp32 autograd gradient is off by a quarter of as much (ill-conditioned roots), and then by
# no more than 4x that.
GRAD_TOL = {torch.float32: 2e-4, torch.float64: 1e-8}
KINK_GAP = {torch.float32: 1e-4, torch.float64: 1e-8}


def _oracle_grad(cs, x, G, dtype=torch.float64, method="RAYEN"):
    buf = oracle.precompute(csd_from_cs(cs), dtype)
    xr = x.to(dtype).clone().requires_grad_(True)
    y = oracle.forward(buf, xr, method=method)
    (y[:, :, 0] * G.to(dtype)).sum().backward()
    return xr.grad[:, :, 0].double().numpy(), y.detach()[:, :, 0].double().numpy()


def _row_err(got, want, floor=None):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    scale = np.maximum(np.max(np.abs(want), axis=1), 1e-12 if floor is None else floor)
    return np.max(np.abs(got - want), axis=1) / scale


def _assert_gradient(got, want, cs, x, G, dtype, method="RAYEN", floor=None, what=""):
    """``got`` against the fp64 truth ``want`` under the rule above."""
    err = _row_err(got, want, floor)
    kink = kink_mask(oracle, cs, x, KINK_GAP[dtype], method=method)
    bound = np.full(err.shape, GRAD_TOL[dtype])
    if dtype == torch.float32:
        try:
            theirs = _row_err(_oracle_grad(cs, x, G, torch.float32, method)[0], want, floor)
            bound = np.maximum(bound, 4.0 * np.nan_to_num(theirs, nan=0.0))
        except AssertionError:          # the reference's fp32 discriminant went negative somewhere
            pass
    # nearly repeated top LMI eigenvalues that are not close enough to be a kink: what a backward-stable eigen-solver
    # delivers at the working precision (helpers.lmi_gradient_bound)
    bound = np.maximum(bound, lmi_gradient_bound(oracle, cs, x, 6e-8 if dtype == torch.float32 else 1.1e-16))
    bad = (~kink) & ~(err <= bound)
    assert not bad.any(), (what, int(bad.sum()), int(kink.sum()), np.flatnonzero(bad)[:5], err[bad][:5], bound[bad][:5])
    assert kink.sum() <= max(2, 0.02 * kink.size), (what, "too many samples classified as kinks", kink.mean())
    assert np.median(err) <= GRAD_TOL[dtype] / 10


@pytest.mark.parametrize("name", ["c1", "c2", "c3", "c4", "c5r", "mixed"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_backward_matches_oracle_autograd(name, dtype):
    raw = _cases()[name]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    B = 512
    gen = torch.Generator().manual_seed(77)
    scale = 3.0 if name == "c1" else 1.0
    x = (torch.empty(B, cs.n, 1).uniform_(-scale, scale, generator=gen)).to(dtype)
    x[:8] *= 1e-3                                            # a few interior (unclipped) samples
    G = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen).to(dtype)

    xg = x.cuda().requires_grad_(True)
    y = layer(xg)
    (y[:, :, 0] * G.cuda()).sum().backward()
    got = xg.grad[:, :, 0].cpu().double().numpy()
    want, y_ref = _oracle_grad(cs, x, G)

    assert np.all(np.isfinite(got))
    _assert_gradient(got, want, cs, x, G, dtype, what=name)
    # interior samples: y = y0 + NA_E v exactly, so grad = NA_E' G
    lift = G[:8].double().numpy() @ cs.NA_E
    assert np.max(np.abs(got[:8] - lift)) <= 1e-5 * max(1.0, np.max(np.abs(lift)))


def test_training_step_through_a_model():
    """loss.backward() + optimiser step through mapper -> projection (examples/main.py:166-171 usage)."""
    cs = workloads.build_constraints(workloads.make_raw("c2", seed=5))
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(),
                                ConstraintModule(cs, input_dim=32, create_map=True)).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    x = torch.randn(256, 6, device="cuda")
    target = torch.zeros(256, cs.k, 1, device="cuda")
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = ((model(x) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]
    for p in model.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "mixed"])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_backward_rayen_old_head(name, dtype):
    """Gradients w.r.t. the direction AND the step column beta of method='RAYEN_old'."""
    raw = _cases()[name]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, method="RAYEN_old", create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    B = 400
    gen = torch.Generator().manual_seed(5)
    x = torch.empty(B, cs.n + 1, 1).uniform_(-1, 1, generator=gen).to(dtype)
    G = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen).to(dtype)
    xg = x.cuda().requires_grad_(True)
    (layer(xg)[:, :, 0] * G.cuda()).sum().backward()
    got = xg.grad[:, :, 0].cpu().double().numpy()

    want, _ = _oracle_grad(cs, x, G, torch.float64, "RAYEN_old")
    assert np.all(np.isfinite(got))
    _assert_gradient(got, want, cs, x, G, dtype, method="RAYEN_old", what=name)


# ---------------------------------------------------------------------------------------------
# matrix-core backward (rayen_mfma_bwd.hip) against the lane-per-sample backward on the same inputs
# ---------------------------------------------------------------------------------------------

def _bwd_sets():
    rng = np.random.default_rng(3)
    k = 40
    lowrank = workloads.random_lin_quad_soc(k=k, m=64, n_quad=0, n_soc=1, r_M=12, seed=61)   # short cone block
    for _ in range(3):                                                                        # rank-12 quadratics
        U = rng.uniform(-1, 1, size=(12, k))
        lowrank["P"].append(U.T @ U)
        lowrank["q"].append(rng.uniform(-1, 1, size=(k, 1)))
        lowrank["r"].append(rng.uniform(-1, 0, size=(1, 1)))
    eq_dense = workloads.corridor_like(k=40, n_eq=10, m=96, n_quad=0, rank=3, seed=62)     # equalities, n = 30
    for _ in range(2):                                                                      # + dense quadratics
        T = rng.uniform(-1, 1, size=(40, 40))
        P = T @ T.T
        q = rng.uniform(-1, 1, size=(40, 1))
        g0 = 0.5 * eq_dense["y0"].T @ P @ eq_dense["y0"] + q.T @ eq_dense["y0"]
        eq_dense["P"].append(P)
        eq_dense["q"].append(q)
        eq_dense["r"].append(-g0 - rng.uniform(0.1, 1.0, size=(1, 1)))
    packed_identity = workloads.random_lin_quad_soc(k=32, m=64, n_quad=0, n_soc=0, seed=63)  # NA_E = I, packed factors
    for r_ in (2, 3, 6, 8, 4, 1, 7, 3, 5, 2):
        U = rng.uniform(-1, 1, size=(r_, 32))
        packed_identity["P"].append(U.T @ U)
        packed_identity["q"].append(rng.uniform(-1, 1, size=(32, 1)))
        packed_identity["r"].append(rng.uniform(-1, 0, size=(1, 1)))
    many_packed = workloads.random_lin_quad_soc(k=32, m=64, n_quad=0, n_soc=0, seed=64)      # NA_E = I, 70 small factors:
    for i in range(70):                                                                       # too many for the dense walk
        U = rng.uniform(-1, 1, size=(1 + i % 5, 32))
        many_packed["P"].append(U.T @ U)
        many_packed["q"].append(rng.uniform(-1, 1, size=(32, 1)))
        many_packed["r"].append(rng.uniform(-1, 0, size=(1, 1)))
    return {
        "eq_packed_n50": workloads.corridor_like(k=60, n_eq=10, m=96, n_quad=20, rank=3, seed=65),   # two blocks of v
        "many_packed": many_packed,
        "c5r": workloads.make_raw("c5r", seed=55),                  # equalities + 72 packed rank-3 quadratics
        "eq_dense": eq_dense,
        "packed_identity": packed_identity,
        "c2": workloads.make_raw("c2", seed=51),
        "c3": workloads.make_raw("c3", seed=52),
        "lin": workloads.random_lin_quad_soc(k=48, m=200, n_quad=0, n_soc=0, seed=53),
        "soc_only": workloads.random_lin_quad_soc(k=32, m=0, n_quad=0, n_soc=3, seed=54),
        "lowrank": lowrank,
    }


@pytest.mark.parametrize("name", ["c2", "c3", "lin", "soc_only", "lowrank", "c5r", "eq_dense", "packed_identity",
                                  "many_packed", "eq_packed_n50"])
@pytest.mark.parametrize("old_head", [False, True])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_matrix_core_backward_matches_lane_backward(name, old_head, dtype):
    from rayen_amd import ops
    cs = workloads.build_constraints(_bwd_sets()[name])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ConstraintModule(cs, create_map=False, method="RAYEN_old" if old_head else "RAYEN").cuda()
    finally:
        torch.set_default_dtype(prev)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    B = 1237                                                  # ragged: not a multiple of 32
    gen = torch.Generator().manual_seed(9)
    width = cs.n + (1 if old_head else 0)
    v = torch.empty(B, width).uniform_(-1.5, 1.5, generator=gen)
    v[:40] *= 1e-3                                            # interior: unclipped
    if not old_head:
        v[40:44] = 0.0                                        # (autograd through v/||v|| is NaN at 0 for RAYEN_old)
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
    v, g = v.to(dtype).cuda(), g.to(dtype).cuda()
    _, kappa, active = ops.project_raw(v, dp, want_active=True, old_head=old_head)
    got = ops.backward_raw(v, kappa, active, g, dp, old_head=old_head)
    assert torch.isfinite(got).all()
    if old_head:
        # no forced-generic entry point for the old head: compare with autograd through the fp64 oracle
        buf = oracle.precompute(csd_from_cs(cs), torch.float64)
        xr = v.cpu().double().unsqueeze(2).requires_grad_(True)
        y = oracle.forward(buf, xr, method="RAYEN_old")
        (y[:, :, 0] * g.cpu().double()).sum().backward()
        want = xr.grad[:, :, 0]
    else:
        want = ops.backward_raw(v, kappa, active, g, dp, force_generic=True).cpu().double()
    got = got.cpu().double()
    method = "RAYEN_old" if old_head else "RAYEN"
    x3, g_cpu = v.cpu().unsqueeze(2), g.cpu()
    if old_head:
        truth = want.numpy()
    else:
        # the lane-per-sample backward reads the same (kappa, active) record, so the two kernels take the same branch
        # at every sample, kinks included: they must agree everywhere to rounding ...
        err = _row_err(got.numpy(), want.numpy())
        assert err.max() <= (2e-4 if dtype == torch.float32 else 1e-9), np.sort(err)[-5:]
        # ... and the matrix-core one is also held to the fp64 autograd truth
        try:
            truth, _ = _oracle_grad(cs, x3.clone(), g_cpu, torch.float64, method)
        except AssertionError:
            # the reference op sequence asserts (NaN) when a ray never meets a cone (negative discriminant, CM:342;
            # the kernels give that cone kappa = 0): no autograd truth for this set, the kernel-vs-kernel check stands
            return
        truth[40:44] = g_cpu[40:44].double().numpy() @ np.asarray(cs.NA_E)      # v = 0: the identity map around 0
    _assert_gradient(got.numpy(), truth, cs, x3, g_cpu, dtype, method=method, what=name)


@pytest.mark.parametrize("name", ["r3_lin", "r7", "r12_lin_eq", "r16", "r24_lin", "r30_eq", "c4"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_quad_lmi_backward_matches_lane_backward_and_oracle(name, dtype):
    """The four-lanes-per-sample LMI backward (eigenvector by inverse iteration on the tridiagonal form) against
    the lane-per-sample Jacobi backward and against autograd through the fp64 oracle's eigvalsh."""
    import importlib.util, os
    from rayen_amd import ops
    from rayen_amd._lib import RayenError
    spec = importlib.util.spec_from_file_location("_parity", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    parity = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(parity)
    raw = workloads.make_raw("c4", seed=33) if name == "c4" else parity._lmi_cases()[name]
    _check_lmi_backward(raw, dtype)


def _random_lmi_set(seed):
    """[linear rows] + [equalities] + one LMI around a random interior point: what the quad kernels serve."""
    rng = np.random.default_rng(5000 + seed)
    k = int(rng.integers(2, 13))
    r = int(rng.choice([2, 3, 5, 8, 9, 13, 16, 17, 20, 23, 24, 27, 32]))
    raw = workloads._empty(k)
    y0 = rng.uniform(-1, 1, size=(k, 1))
    raw["y0"] = y0
    n_eq = int(rng.integers(0, min(3, k - 1) + 1)) if rng.random() < 0.4 else 0
    if n_eq:
        raw["A2"] = rng.uniform(-1, 1, size=(n_eq, k))
        raw["b2"] = raw["A2"] @ y0
    m = int(rng.choice([0, 0, 5, 40]))
    if m:
        raw["A1"] = rng.uniform(-1, 1, size=(m, k))
        raw["b1"] = raw["A1"] @ y0 + rng.uniform(0.1, 1.0, size=(m, 1))
    F = []
    for _ in range(k):
        T = rng.uniform(-1, 1, size=(r, r))
        F.append((T + T.T) / 2)
    T = rng.uniform(-1, 1, size=(r, r))
    H = T @ T.T + 0.5 * np.eye(r)
    F.append(H - sum(y0[i, 0] * F[i] for i in range(k)))
    raw["F"] = F
    return raw


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("RAYEN_FUZZ_SEEDS", "100")) // 8)))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_random_lmi_sets_backward(seed, dtype):
    _check_lmi_backward(_random_lmi_set(seed), dtype)


def _check_lmi_backward(raw, dtype):
    from rayen_amd import ops
    from rayen_amd._lib import RayenError
    cs = workloads.build_constraints(raw)
    r = cs.lmic.all_F[0].shape[0]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    B = 777
    gen = torch.Generator().manual_seed(19)
    v = torch.empty(B, cs.n).uniform_(-2.0, 2.0, generator=gen)
    v[:20] *= 1e-3
    v[20:22] = 0.0
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
    vd, gd = v.to(dtype).cuda(), g.to(dtype).cuda()
    try:
        y_dev, kappa, active = ops.project_raw(vd, dp, want_active=True)
    except RayenError:
        assert dtype == torch.float64 and r > 24               # no fp64 forward at all for this size
        return
    try:
        got = ops.backward_raw(vd, kappa, active, gd, dp).cpu().double()
    except RayenError:
        assert dtype == torch.float64 and r > 20               # neither the quad nor the lane kernel fits
        return
    assert torch.isfinite(got).all()

    buf = oracle.precompute(csd_from_cs(cs), torch.float64)
    xr = v.double().unsqueeze(2).requires_grad_(True)
    y = oracle.forward(buf, xr)
    (y[:, :, 0] * g.double()).sum().backward()
    y_true = y.detach()[:, :, 0]
    y_err = (y_dev.cpu().double() - y_true).abs().amax(1) / y_true.abs().amax(1).clamp_min(1e-30)
    if dtype == torch.float32:
        # forward parity of the LMI kernels: the north_star's 1e-5, or twice what LAPACK's own fp32 eigvalsh (the
        # reference's arithmetic) leaves against the fp64 truth on the same inputs
        y32 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), v.unsqueeze(2))[:, :, 0].double()
        theirs = float(((y32 - y_true).abs().amax(1) / y_true.abs().amax(1).clamp_min(1e-30)).max())
        assert float(y_err.max()) <= max(1e-5, 2.0 * theirs), (float(y_err.max()), theirs, r)
    else:
        assert float(y_err.max()) <= 1e-9
    want = xr.grad[:, :, 0]
    # v = 0: eigvalsh of the zero matrix has no autograd derivative worth comparing; the layer is the identity
    # map around 0, so its gradient there is NA_E' g
    want[20:22] = g[20:22].double() @ torch.from_numpy(np.asarray(cs.NA_E, dtype=np.float64))
    # (a clipped sample of a one-dimensional set has gradient exactly 0 -- y does not move with v --: errors are
    # measured against the incoming gradient's size there, not against rounding noise)
    # the two terms that cancel are of the size of g; rounding leaves a few ulps of THAT (observed <= 5e-16 |g| in
    # fp64 over 250 sets), so the floor is |g| * 1e-6: an absolute error of 1e-14 |g| still fails
    # fp32 (round 4, 1000-seed fuzz; round 5: the floor is 1e-3 |g| again for every set): on ONE- and TWO-dimensional
    # feasible sets a clipped sample's true gradient is (nearly) zero, and what the kernel returns there is the residue of
    # that cancellation, 3e-7 ... 5e-7 |g| (2 - 4 ulps of the two terms; the forward's kappa is within 2e-7 of the fp64
    # oracle's, scripts/ubench/lmi_kappa_check.py).  Those rows -- n <= 2 and |true gradient| < 1e-3 |g| -- are held to an
    # explicit ABSOLUTE bound, 1e-6 |g|, instead of a relative one against a number that is zero; every other row of
    # every set keeps the relative bar with the 1e-3 |g| floor.
    gmag = g.double().abs().amax(1)
    floor = gmag * (1e-3 if dtype == torch.float32 else 1e-6)
    if dtype == torch.float32 and cs.n <= 2:
        tiny = want.abs().amax(1) < 1e-3 * gmag
        if bool(tiny.any()):
            resid = (got[tiny].double() - want[tiny]).abs().amax(1)
            assert bool((resid <= 1e-6 * gmag[tiny]).all()), (float((resid / gmag[tiny]).max()), r, cs.n)
            got = got.clone()
            got[tiny] = want[tiny].to(got.dtype)
    _assert_gradient(got.numpy(), want.numpy(), cs, v.unsqueeze(2), g, dtype, floor=floor.numpy(), what=f"lmi r={r}")
    err = torch.from_numpy(_row_err(got.numpy(), want.numpy(), floor.numpy()))
    assert float(err[:22].max()) <= (1e-5 if dtype == torch.float32 else 1e-12)

    try:
        lane = ops.backward_raw(vd, kappa, active, gd, dp, force_generic=True).cpu().double()
    except RayenError:                                          # the lane-per-sample kernel stops at ~21 x 21 (fp64)
        assert r > 20
        return
    _assert_gradient(lane.numpy(), want.numpy(), cs, v.unsqueeze(2), g, dtype, floor=floor.numpy(), what=f"lmi lane r={r}")


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("RAYEN_FUZZ_SEEDS", "100")) // 3)))
def test_random_sets_gradients_match_oracle_autograd(seed):
    """fp64 gradients of random constraint sets (whatever backward kernel serves them) against autograd through
    the fp64 oracle; both heads."""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("_parity", os.path.join(os.path.dirname(__file__), "test_gpu_parity.py"))
    parity = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(parity)
    raw = parity._random_set(9000 + seed)
    rng = np.random.default_rng(seed)
    method = "RAYEN_old" if rng.random() < 0.3 else "RAYEN"
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, create_map=False, method=method).cuda()
    finally:
        torch.set_default_dtype(prev)
    B = int(rng.choice([17, 64, 500]))
    gen = torch.Generator().manual_seed(seed)
    width = layer.getDimAfterMap()
    x = torch.empty(B, width, 1, dtype=torch.float64).uniform_(-1.5, 1.5, generator=gen)
    G = torch.empty(B, cs.k, dtype=torch.float64).uniform_(-1, 1, generator=gen)
    buf = oracle.precompute(csd_from_cs(cs), torch.float64)
    xr = x.clone().requires_grad_(True)
    try:
        y_ref = oracle.forward(buf, xr, method=method)
    except AssertionError:
        pytest.skip("the reference asserts on this set (a ray that never meets a cone)")
    (y_ref[:, :, 0] * G).sum().backward()
    want = xr.grad[:, :, 0]
    xg = x.cuda().requires_grad_(True)
    (layer(xg)[:, :, 0] * G.cuda()).sum().backward()
    got = xg.grad[:, :, 0].cpu()
    assert torch.isfinite(got).all()
    _assert_gradient(got.numpy(), want.numpy(), cs, x, G, torch.float64, method=method, what=f"seed {seed}")


@pytest.mark.parametrize("name,B,dtype", [
    ("c3", 70001, torch.float32), ("eq_free_2quad", 40000, torch.float32), ("lowrank", 33000, torch.float32),
    ("soc_quad", 50000, torch.float32),
    ("c3", 70001, torch.float64), ("eq_free_2quad", 40000, torch.float64), ("lowrank", 33000, torch.float64),
    ("soc_quad", 50000, torch.float64),
    # the general fp32 backward (equality constraints / packed low-rank quadratics): buckets = packed tile pairs + dense forms
    ("c5r", 66000, torch.float32), ("eq_packed_n50", 40000, torch.float32), ("many_packed", 50000, torch.float32),
    ("packed_identity", 33333, torch.float32)])
def test_bucketed_backward_equals_the_plain_walk(name, B, dtype):
    """Large batches of packs with several dense forms are grouped by active constraint first (three small
    launches in a scratch buffer) and every group walks only its own form: same bits as the walk over every form,
    whatever mix of buckets (not clipped / linear row / each form) a batch holds."""
    from rayen_amd import _lib, ops
    extra = {"eq_free_2quad": workloads.random_lin_quad_soc(k=64, m=128, n_quad=2, n_soc=0, seed=66),   # 2 forms x 2 tiles
             "soc_quad": workloads.random_lin_quad_soc(k=40, m=20, n_quad=1, n_soc=2, seed=67)}
    cs = workloads.build_constraints(extra[name] if name in extra else _bwd_sets()[name])
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    dp_layer = dp
    if dtype == torch.float32 and dp.info().bwd_f32 in (3, 7):
        # (the f16-pair backward streams the batch in order and asks for no workspace: the bucketed walk of the exact-fp32
        # kernel behind it is what this test is about)
        from rayen_amd import pack as _pack
        dp = _pack.DevicePack(layer.packed_constants(), 0, fp32_mode=1)
    query = getattr(_lib.load(), "rayen_bwd_workspace_bytes_" + ("f32" if dtype == torch.float32 else "f64"))
    assert int(query(dp.handle, B)) > 0, "the pack should take the bucketed walk"
    assert int(query(dp.handle, 1000)) == 0                 # small batches: plain walk
    gen = torch.Generator(device="cuda").manual_seed(5)
    v = torch.empty(B, cs.n, device="cuda", dtype=dtype).uniform_(-1.5, 1.5, generator=gen)
    v[: B // 5] *= 0.05                                     # a fifth of the batch stays inside the set (bucket 0)
    v[B // 5: B // 5 + 3] = 0.0
    g = torch.empty(B, cs.k, device="cuda", dtype=dtype).uniform_(-1, 1, generator=gen)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    assert 0.05 < float((kappa > 1).double().mean()) < 1.0
    want = ops.backward_raw(v, kappa, active, g, dp, bucketed=False)
    for _ in range(2):                                      # (the scratch buffer is re-initialised by every call)
        got = ops.backward_raw(v, kappa, active, g, dp, bucketed=True)
        assert torch.equal(got, want)
    # through autograd (the registered op takes the bucketed path by itself)
    xg = v.clone().unsqueeze(2).requires_grad_(True)
    (layer(xg)[:, :, 0] * g).sum().backward()
    if dp_layer is not dp:                                  # (the module's own pack: its record, its backward kernel)
        _, kappa, active = ops.project_raw(v, dp_layer, want_active=True)
        want = ops.backward_raw(v, kappa, active, g, dp_layer)
    assert torch.equal(xg.grad[:, :, 0], want)


def test_bucketed_walk_of_the_last_form_of_an_even_item_list():
    """n <= 32 (one tile per form), four dense quadratics, NA_E = I: the bucket of the last form walks [it_lo, it_hi) with
    a look-ahead of two tiles, i.e. it touches two tiles behind the list -- they exist since round 3 (ADVICE round 2:
    one spare tile was one too few; a read past a hipMalloc allocation can fault).  Bucketed and plain walks agree bit
    for bit and with the lane-per-sample backward."""
    from rayen_amd import _lib, ops
    raw = workloads.random_lin_quad_soc(k=32, m=48, n_quad=4, n_soc=0, seed=91)
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    assert dp.info().bwd_f32 == 1
    B = 40000
    assert int(_lib.load().rayen_bwd_workspace_bytes_f32(dp.handle, B)) > 0
    gen = torch.Generator(device="cuda").manual_seed(4)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1, generator=gen)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    segs = active[:, 0][kappa > 1]
    assert int((segs == segs.max()).sum()) > 0                      # the last form is somebody's active constraint
    want = ops.backward_raw(v, kappa, active, g, dp, bucketed=False)
    for _ in range(3):
        assert torch.equal(ops.backward_raw(v, kappa, active, g, dp, bucketed=True), want)
    lane = ops.backward_raw(v, kappa, active, g, dp, force_generic=True)
    size = lane.abs().amax(1).clamp_min(1e-20)
    assert float(((want - lane).abs().amax(1) / size).max()) <= 2e-4
