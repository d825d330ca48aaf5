"""Parity tests proper: the HIP path (through the C ABI) against the reference's golden vectors,
the CPU oracle, closed-form answers and size-independent properties.  Needs an MI355X."""
import os

import numpy as np
import pytest
import torch

from helpers import golden_names, load_golden, rel_err_rows, csd_from_cs, raw_from_cs
from oracle import rayen_oracle as oracle
from rayen_amd import constraints, ops, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5     # BASELINE.json: "within 1e-5 relative fp32" (per-sample inf-norm relative error)
FP64_TOL = 1e-9
VIOLATION_TOL = 1e-6


def _layer(raw_or_cs, dtype=torch.float32, **kw):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = raw_or_cs if isinstance(raw_or_cs, constraints.ConvexConstraints) \
            else workloads.build_constraints(raw_or_cs)
        layer = ConstraintModule(cs, method="RAYEN", create_map=False, **kw).to("cuda")
        layer.register_forward_hook(_served_by_a_kernel)
        return cs, layer
    finally:
        torch.set_default_dtype(prev)


def _served_by_a_kernel(module, args, output):
    """Every forward of a parity test ran on a hand-written kernel: the module never switched to the packed torch
    evaluator on the device libraries (rayen_amd/eager.py).  tests/conftest.py already makes that switch an error
    (RAYEN_STRICT_HIP=1, the announcement a raised warning); this is the belt to those braces."""
    assert not module._hip_unsupported, "the HIP kernels refused this set; the answer came from the device-library detour"


def test_a_kernel_that_refuses_a_baseline_shape_fails_this_suite(monkeypatch):
    """The guard itself: make the C ABI refuse config 3 (RAYEN_E_UNSUPPORTED from every forward entry point).  Under
    this suite's environment the module raises; with the environment lifted the detour answers (correctly, loudly) and
    the forward hook of ``_layer`` fails the test."""
    import warnings
    from rayen_amd import _lib
    cs, layer = _layer(workloads.make_raw("c3", seed=0))
    x = torch.empty(256, cs.n, 1).uniform_(-1, 1).cuda()
    good = layer(x)
    assert _lib.load().rayen_last_forward_kernel() != _lib.KERNEL_NONE
    for name in list(ops._FWD.values()) + list(ops._FWD_OLD.values()):
        monkeypatch.setitem(ops._ENTRY, name, lambda *a, **k: _lib.E_UNSUPPORTED)
    layer._fast.clear()
    with pytest.raises(_lib.RayenError):
        layer(x)
    assert not layer._hip_unsupported
    monkeypatch.delenv("RAYEN_STRICT_HIP")
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with pytest.raises(AssertionError, match="device-library detour"):
            layer(x)
        layer._forward_hooks.clear()
        detour = layer(x)
    assert any("no HIP kernel serves" in str(w.message) for w in caught)
    assert float((detour - good).abs().max()) <= 1e-5      # (the detour's answer is right: that is why it must be loud)


def _to_my_basis(cs, csd_ref, x, dtype):
    R = cs.NA_E.T @ csd_ref["NA_E"]
    v = x[:, :, 0].astype(np.float64) @ R.T
    return torch.tensor(v, dtype=dtype).unsqueeze(2)


def _packed_truth(cs, v64):
    """(y, kappa) of the fp64 packed form (tests/packed_eval.py, pinned to the reference's fp64 outputs on every golden
    fixture) -- the truth where the reference's own op sequence is NaN."""
    import packed_eval
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        consts = ConstraintModule(cs, method="RAYEN", create_map=False).packed_constants()
    finally:
        torch.set_default_dtype(prev)
    y, kappa, _ = packed_eval.evaluate(consts, np.asarray(v64, dtype=np.float64))
    return y, kappa


SUBSTITUTED_CAP = 0.05      # at most this fraction of a batch may take the packed form as its truth
BISECTED_ROWS = 1024        # rows of every replaced batch held to bisection on the raw constraints


_ORACLE_CACHE = {}


def _oracle_forward(cs, x_cpu, dtype):
    """Memoised per (constraint data, inputs, dtype): several tests of this file hold different kernels to the SAME oracle
    outputs, and on config 5 one evaluation (with its bisection of the replaced rows) takes half a minute of host time."""
    import hashlib
    key = (str(dtype), tuple(x_cpu.shape), hashlib.sha1(x_cpu.contiguous().numpy().tobytes()).hexdigest(),
           hashlib.sha1(np.ascontiguousarray(np.asarray(cs.A_p, dtype=np.float64)).tobytes()
                        + np.ascontiguousarray(np.asarray(cs.y0, dtype=np.float64)).tobytes()).hexdigest(),
           len(cs.qcs), len(cs.socs), bool(cs.has_lmi_constraints))
    if key not in _ORACLE_CACHE:
        if len(_ORACLE_CACHE) > 64:
            _ORACLE_CACHE.clear()
        _ORACLE_CACHE[key] = _oracle_forward_uncached(cs, x_cpu, dtype)
    return _ORACLE_CACHE[key].copy()


def _oracle_forward_uncached(cs, x_cpu, dtype):
    """The reference's op sequence at ``dtype`` -- and where THAT is NaN, the reference's op sequence at fp64, rounded.

    On the corridor set (config 5) the reference takes ``sqrt`` of slightly negative radicands (CM:374): NaN on EVERY row at
    fp32 on the GPU boxes, on ~1 % at fp64.  Round 5 replaced all of those rows by the product's own
    packed formulation (tests/packed_eval.py), i.e. compared the kernel with itself (round-5 verdict, weak 1).  Now:
      1. rows that are NaN at ``dtype`` take the fp64 ORACLE (same reference op sequence, CM:351-474) rounded to ``dtype``
         -- exactly how ``helpers.load_golden`` treats the reference's own golden outputs;
      2. only rows on which the fp64 op sequence is NaN too take the fp64 packed form (normally ~1 % of the batch; the
         fraction is printed against SUBSTITUTED_CAP);
      3. replaced rows (both kinds; BISECTED_ROWS of them, every packed-form row included -- ALL packed-form rows when
         their fraction is beyond the cap) are held to the step along the ray found by bisection on the RAW constraints
         in fp64 -- shares nothing with W, the packed constants or any kappa formula.
    Any other NaN (a set with cones or an LMI, NaN inputs) asserts as before (CM:531)."""
    y = oracle.forward(oracle.precompute(csd_from_cs(cs), dtype), x_cpu.to(dtype), check_nan=False).numpy()[:, :, 0]
    bad = ~np.isfinite(y).all(axis=1)
    if bad.any():
        assert len(cs.qcs) and not len(cs.socs) and not cs.has_lmi_constraints and np.isfinite(x_cpu.numpy()).all()
        rows = np.flatnonzero(bad)
        x64 = x_cpu[rows].to(dtype).double()                  # (the inputs as the dtype run saw them)
        y64 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x64, check_nan=False).numpy()[:, :, 0]
        bad64 = ~np.isfinite(y64).all(axis=1)
        truth = y64
        if bad64.any():
            packed, _ = _packed_truth(cs, x64[bad64][:, :cs.n, 0].numpy())
            truth = y64.copy()
            truth[bad64] = packed
        # which rows the fp64 op sequence loses is the HOST's arithmetic (1 % of config 5's rows on most boxes, 70 % seen on
        # one), so the cap cannot be a hard failure of the product's suite: within the cap BISECTED_ROWS rows
        # are bisected (every packed-form row among them); beyond it EVERY packed-form row is -- nothing the packed form
        # says is then taken on trust
        over_cap = bad64.sum() > SUBSTITUTED_CAP * len(bad)
        pick = np.random.default_rng(0).permutation(len(rows))[:BISECTED_ROWS]
        pick = np.unique(np.concatenate((np.flatnonzero(bad64)[:None if over_cap else BISECTED_ROWS], pick)))
        indep = _ray_bisection_truth(cs, x64[pick][:, :cs.n, 0].numpy())
        gap = rel_err_rows(truth[pick], indep)
        # (the fp64 op sequence takes the square root of a radicand that cancels to ~1e-16 of its terms: where such a form is the
        # active one its kappa carries ~1e-8, seen up to 4.5e-8 on one GPU box's host -- two orders inside the 1e-5 bar this truth backs)
        assert gap.max() <= 5e-7, ("replacement rows against bisection on the raw constraints", gap.max())
        print(f"\n  [oracle NaN rows] {len(rows)} / {len(bad)} at {str(dtype)[6:]} take the fp64 ORACLE rounded; of those "
              f"{int(bad64.sum())} ({bad64.sum() / len(bad):.2%} of the batch, cap {SUBSTITUTED_CAP:.0%}"
              f"{': EXCEEDED on this host, every such row bisected' if over_cap else ''}) are NaN at fp64 too and "
              f"take the packed form; {len(pick)} replaced rows against bisection on the raw constraints: {gap.max():.2e}")
        y[rows] = truth.astype(y.dtype)
    return y


def _ray_bisection_truth(cs, v64):
    """y = y0 + t NA_E v with t = min(1, sup{t : y0 + t NA_E v feasible}) by bisection on the raw constraints' residuals
    (ConvexConstraints.getResiduals, fp64; the equalities hold along the whole ray).  Independent of W, of the packed
    constants and of every kappa formula."""
    d = np.asarray(v64, dtype=np.float64) @ np.asarray(cs.NA_E, dtype=np.float64).T
    y0 = np.asarray(cs.y0, dtype=np.float64)[:, 0]

    def worst(y):
        res = cs.getResiduals(y)
        return np.max(np.stack([val for key, val in res.items() if key != "lin_eq"], axis=0), axis=0)

    inside = worst(y0[None] + d) <= 0.0
    lo, hi = np.zeros(len(d)), np.ones(len(d))
    for _ in range(60):                       # (halving until a feasible step is known -- t may be 1e-12 --, then the gap
                                              # between lo and hi halves in the LOGARITHM every step: 1e-15 long before 60)
        mid = np.where(lo > 0.0, np.sqrt(lo * hi), 0.5 * hi)
        ok = worst(y0[None] + mid[:, None] * d) <= 0.0
        lo, hi = np.where(ok, mid, lo), np.where(ok, hi, mid)
    t = np.where(inside, 1.0, lo)
    return y0[None] + t[:, None] * d


# --------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("tag,dtype,tol", [("32", torch.float32, FP32_TOL), ("64", torch.float64, FP64_TOL)])
def test_golden(name, tag, dtype, tol):
    raw, csd, z = load_golden(name)
    cs, layer = _layer(raw, dtype)
    x = _to_my_basis(cs, csd, z["x"], dtype)
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    assert y.shape == z["y" + tag].shape
    assert np.max(rel_err_rows(y, z["y" + tag])) <= tol
    # no worse than the reference's own output at this precision (its fp32 y is not bit-exactly
    # feasible either), and below 1e-6 wherever the reference is
    ref_violation = oracle.max_violation(raw, z["y" + tag])
    floor = VIOLATION_TOL if tag == "32" else 1e-11
    assert oracle.max_violation(raw, y) <= max(floor, 3 * ref_violation)
    # the computeKappa helper on normalised directions
    v_bar = torch.nn.functional.normalize(x[:, 0:cs.n, 0:1], dim=1)
    kb = layer.computeKappa(v_bar.cuda()).cpu().numpy()[:, 0, 0]
    ref = z["kappa_bar" + tag]
    assert kb.shape == ref.shape
    # (kappa of a unit direction, relative to max(1, kappa): the same quantity the 1e-5 bar on y measures once the
    # step is clipped; fp64: the reference's own kappa carries sqrt/eigvalsh rounding ~1e-13 x its magnitude)
    assert np.max(np.abs(kb - ref) / np.maximum(1.0, np.abs(ref))) <= (2 * tol if tag == "32" else 1e-8)


@pytest.mark.parametrize("name", golden_names())
def test_golden_generic_fp32_kernel(name):
    """Same vectors through the generic (lane = sample) fp32 kernel, whatever the dispatcher prefers."""
    raw, csd, z = load_golden(name)
    cs, layer = _layer(raw, torch.float32)
    x = _to_my_basis(cs, csd, z["x"], torch.float32).cuda()
    dp, _ = layer.device_pack(x.device)
    y, kappa, active = ops.project_raw(x.reshape(x.shape[0], -1), dp, force_generic=True)
    assert np.max(rel_err_rows(y.cpu().numpy(), z["y32"])) <= FP32_TOL
    assert bool((kappa >= 0).all())
    a = active.cpu().numpy()
    assert np.all((a[:, 0] >= -1) & (a[:, 0] < len(dp.consts.segments)))
    assert np.all((a[:, 0] == -1) == (kappa.cpu().numpy() == 0))


def _violation_bound(raw, cs, v, dtype):
    """The north_star's 1e-6 (absolute, unnormalised residuals), or three times the violation of the reference's
    own output at this precision on the same directions ``v [b, n]`` (its fp32 output is not bit-exactly
    feasible either; sets with |P| ~ 1e2 turn one fp32 ulp of y into 1e-5 of residual)."""
    floor = VIOLATION_TOL if dtype == torch.float32 else 1e-11
    y_ref = _oracle_forward(cs, v.detach().cpu().reshape(v.shape[0], -1, 1), dtype)
    return max(floor, 3.0 * oracle.max_violation(raw, y_ref))


def _fp32_bound(cs, x, y_true, layer=None, method="RAYEN", reference_yardstick=True):
    """The fp32 parity bar with its yardsticks, all measured against the fp64 truth ``y_true`` on the same inputs:
    the north_star's 1e-5, or twice the error of the reference's own fp32 arithmetic (oracle at fp32), or -- for a
    set whose constants do not survive fp32 rounding -- four times the error that rounding ALONE causes (the packed
    fp32 constants evaluated in fp64 arithmetic, tests/packed_eval.py)."""
    bound = FP32_TOL
    BOUND_LOG.append(None)           # (filled in below: every bar that was ever applied is on record)
    try:
        # (reference_yardstick=False: inputs on which the reference's fp32 arithmetic overflows -- |v| ~ 1e12 -- would
        # make this yardstick meaningless)
        if reference_yardstick:
            y32 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x.float(), method=method).numpy()[:, :, 0]
            bound = max(bound, 2.0 * rel_err_rows(y32, y_true).max())
    except AssertionError:          # the reference's fp32 discriminant went negative (CM:342)
        pass
    if layer is not None and method == "RAYEN":
        import packed_eval
        y_const, _, _ = packed_eval.evaluate(layer.packed_constants(), x[:, :cs.n, 0].double().numpy())
        bound = max(bound, 4.0 * rel_err_rows(y_const, y_true).max())
    BOUND_LOG[-1] = float(bound)
    BOUND_WHO.append((os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], float(bound)))
    return bound


BOUND_LOG = []
BOUND_WHO = []   # (test id, bar) of every call


@pytest.mark.parametrize("name,B", [("c1", 500), ("c2", 4096), ("c3", 8192), ("c4", 4096), ("c5", 8192), ("c5r", 8192)])
def test_fp32_bar_on_the_baseline_configs_is_the_north_star_itself(name, B, capsys):
    """``_fp32_bound`` lets a set exceed 1e-5 where the reference's own fp32 arithmetic (or the fp32 rounding of its
    constants) does.  On BASELINE.json's five configurations no yardstick is in play: the bar IS 1e-5, and the
    ratio ours / bar is printed (and recorded in gpurun_out/fp32_bound_log.json by the last test of this file)."""
    raw = workloads.make_raw(name, seed=33)
    cs, layer = _layer(raw, torch.float32)
    gen = torch.Generator().manual_seed(9)
    rng = workloads.CONFIGS[name][3]
    x = torch.empty(B, cs.n, 1).uniform_(-rng, rng, generator=gen)
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_true = _oracle_forward(cs, x.double(), torch.float64)
    bound = _fp32_bound(cs, x, y_true, layer)
    worst = float(rel_err_rows(y, y_true).max())
    RATIO_LOG[name] = {"ours": worst, "bar": bound, "ratio": worst / bound}
    with capsys.disabled():
        print(f"\n  [fp32 bar] {name}: ours {worst:.3e} / bar {bound:.3e} = {worst / bound:.3f}")
    assert bound == FP32_TOL, (name, bound)
    assert worst <= bound


RATIO_LOG = {}


# --------------------------------------------------------------------------- oracle on fresh seeds
@pytest.mark.parametrize("name,B", [("c1", 500), ("c2", 4096), ("c3", 4096), ("c4", 2048), ("c5", 4096), ("c5r", 4096)])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.float64, FP64_TOL)])
def test_against_oracle(name, B, dtype, tol):
    raw = workloads.make_raw(name, seed=21)
    cs, layer = _layer(raw, dtype)
    gen = torch.Generator().manual_seed(5)
    rng = workloads.CONFIGS[name][3]
    x = torch.empty(B, cs.n, 1, dtype=torch.float32).uniform_(-rng, rng, generator=gen).to(dtype)
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref = _oracle_forward(cs, x, dtype)
    assert np.max(rel_err_rows(y, y_ref)) <= tol
    floor = VIOLATION_TOL if dtype == torch.float32 else 1e-11
    assert oracle.max_violation(raw, y) <= max(floor, 3 * oracle.max_violation(raw, y_ref))


def _wide_cases():
    """Shapes that exercise the wider MFMA instantiations (n_pad = 96, 128) and non-identity outputs."""
    rng = np.random.default_rng(40)
    a = workloads.random_lin_quad_soc(k=96, m=200, n_quad=3, n_soc=2, seed=41)
    b = workloads.random_lin_quad_soc(k=128, m=64, n_quad=2, n_soc=1, r_M=40, seed=42)
    c = workloads.random_lin_quad_soc(k=100, m=150, n_quad=2, n_soc=2, r_M=100, seed=43)   # + equalities: n = 70
    c["A2"] = rng.uniform(-1, 1, size=(30, 100))
    c["b2"] = np.zeros((30, 1))
    d = workloads.corridor_like(k=80, n_eq=10, m=100, n_quad=40, rank=6, seed=44)        # rank-7 factors: quad pairs
    return {"k96": a, "k128": b, "k100_n70": c, "k80_pairs": d}


@pytest.mark.parametrize("name", ["k96", "k128", "k100_n70", "k80_pairs"])
def test_wide_shapes_against_oracle(name):
    raw = _wide_cases()[name]
    cs, layer = _layer(raw, torch.float32)
    gen = torch.Generator().manual_seed(6)
    x = torch.empty(3000, cs.n, 1).uniform_(-1, 1, generator=gen)
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref = _oracle_forward(cs, x, torch.float32)
    assert np.max(rel_err_rows(y, y_ref)) <= FP32_TOL
    assert oracle.max_violation(raw, y) <= max(VIOLATION_TOL, 3 * oracle.max_violation(raw, y_ref))
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    yg, _, _ = ops.project_raw(x.reshape(3000, -1).cuda(), dp, force_generic=True)
    assert np.max(rel_err_rows(yg.cpu().numpy(), y_ref)) <= FP32_TOL
    assert dp.info().mfma_f32 == 1


FAMILIES = {"exact": ("1", (0, 1)), "triple": ("2", (2,)), "pair": ("3", (3,))}


@pytest.mark.parametrize("family", ["exact", "triple", "pair"])
@pytest.mark.parametrize("name", ["c2", "c3", "c5", "c5r", "served0", "served1", "served2"])
def test_every_fp32_mfma_family(name, family, monkeypatch):
    """RAYEN_FP32_MODE (read when a pack is created; RayenPackDesc.fp32_mode) pins the family that serves the fp32
    forward: 1 the exact-fp32 MFMA kernels (which otherwise serve only n > 64, the RAYEN_old head and their fused
    mapper), 2 the bf16-triple kernel, 3 the f16-pair kernel (the last two without the creation-time measurement).
    Each against the fp64 truth, with the reference's own fp32 arithmetic on the same inputs as yardstick."""
    mode, served = FAMILIES[family]
    # random sets: the first ones the default dispatch puts on the f16 pairs (laid out for the split-operand kernels:
    # every pinned family can serve them -- nothing is skipped)
    raw = _served_random_set(int(name[6:]), 3) if name.startswith("served") else workloads.make_raw(name, seed=21)
    monkeypatch.setenv("RAYEN_FP32_MODE", mode)
    cs, layer = _layer(raw, torch.float32)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    assert dp.info().mfma_f32 in served
    gen = torch.Generator().manual_seed(8)
    x = torch.empty(3001, cs.n, 1).uniform_(-1.5, 1.5, generator=gen)
    x[:8] *= 1e-4
    x[8:16] *= 300.0
    x[16] = 0.0
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    monkeypatch.delenv("RAYEN_FP32_MODE")
    _, layer_default = _layer(cs, torch.float32)
    y_default = layer_default(x.cuda()).cpu().numpy()[:, :, 0]
    try:
        y_true = _oracle_forward(cs, x.double(), torch.float64)
    except AssertionError:
        # the reference op sequence asserts (NaN) when a ray never meets a cone (CM:342): only feasibility can be checked
        assert _relative_violation(raw, y) <= 1e-5 and _relative_violation(raw, y_default) <= 1e-5
        return
    bound = _fp32_bound(cs, x, y_true, layer)
    assert rel_err_rows(y, y_true).max() <= bound, (rel_err_rows(y, y_true).max(), bound)
    assert rel_err_rows(y_default, y_true).max() <= bound, (rel_err_rows(y_default, y_true).max(), bound)
    # (feasibility next to the reference's own fp32 arithmetic: the equality rows of config 5 leave ~2e-6 in fp32)
    try:
        y_ref = _oracle_forward(cs, x, torch.float32)
    except AssertionError:                                   # the reference's fp32 discriminant went negative (CM:342)
        y_ref = y_true
    assert oracle.max_violation(raw, y) <= max(VIOLATION_TOL, 3 * oracle.max_violation(raw, y_ref))


@pytest.mark.parametrize("name", ["c2", "c3", "c5", "c5r"])
def test_pair_kernel_scales_every_row_on_its_own(name):
    """The f16-pair kernel moves every direction into f16 range with its own power of two: rows of very different
    magnitudes in one batch, components spread over many binades inside a row, zero rows, and non-finite rows (which
    must come out non-finite and raise the NaN flag, never contaminate their neighbours)."""
    raw = workloads.make_raw(name, seed=77)
    cs, layer = _layer(raw, torch.float32)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    assert dp.info().mfma_f32 == 3
    gen = torch.Generator().manual_seed(12)
    B = 2048
    x = torch.empty(B, cs.n).uniform_(-1.0, 1.0, generator=gen)
    mag = 10.0 ** torch.randint(-12, 13, (B, 1), generator=gen).float()            # 1e-12 .. 1e12 per row
    x = x * mag
    spread = 2.0 ** (-torch.randint(0, 20, (B // 2, cs.n), generator=gen).float())  # 2^0 .. 2^-19 inside a row
    x[: B // 2] *= spread
    x[5] = 0.0
    y_true = _oracle_forward(cs, x.double().unsqueeze(2), torch.float64)
    y, kappa, _ = ops.project_raw(x.cuda(), dp)
    # (rows with ||v|| < 1e-12: the reference divides by max(||v||, 1e-12) -- F.normalize's eps, CM:469 -- and so returns
    # y0 + v ||v|| / 1e-12 instead of y0 + v; the kernels have no such floor (documented deviation).  Those
    # rows are held to y0 + v itself, every other row to the fp64 truth.)
    tiny = (x.double().norm(dim=1) < 1e-11).numpy()
    assert 0 < tiny.sum() < B // 4
    big = ~tiny
    err = rel_err_rows(y.cpu().numpy()[big], y_true[big])
    bound = _fp32_bound(cs, x[big].unsqueeze(2), y_true[big], layer, reference_yardstick=False)
    assert bound <= 1e-4, bound                                  # (the yardstick itself must stay meaningful)
    assert err.max() <= bound, (err.max(), bound, int(err.argmax()))
    plain = (torch.as_tensor(cs.y0[:, 0])[None, :] + x.double()[tiny] @ torch.as_tensor(np.asarray(cs.NA_E)).T).numpy()
    assert np.max(rel_err_rows(y.cpu().double().numpy()[tiny], plain)) <= 1e-6
    assert np.allclose(y[5].cpu().numpy(), cs.y0[:, 0], atol=1e-6)
    k_true = oracle.compute_kappa(oracle.precompute(csd_from_cs(cs), torch.float64), x.double().unsqueeze(2))[:, 0, 0].numpy()
    if not np.isfinite(k_true).all():       # (config 5: the reference's radicand hazard, see _oracle_forward)
        k_true = np.where(np.isfinite(k_true), k_true, _packed_truth(cs, x.double().numpy())[1])
    # (kappa against the truth, with what the fp32 rounding of the constants alone does to it as yardstick: config 5)
    import packed_eval
    _, k_const, _ = packed_eval.evaluate(layer.packed_constants(), x.double().numpy())
    size = np.maximum(np.abs(k_true), 1e-30)
    k_bound = max(1e-5, 4.0 * float(np.max(np.abs(k_const - k_true) / size)))
    assert np.all(np.abs(kappa.cpu().numpy() - k_true) <= k_bound * size), (k_bound, float(np.max(np.abs(kappa.cpu().numpy() - k_true) / size)))
    # non-finite rows
    xb = x.clone()
    xb[7, 3] = float("nan")
    xb[9, 0] = float("inf")
    dp.nan_flag.zero_()
    yb, _, _ = ops.project_raw(xb.cuda(), dp)
    yb = yb.cpu().numpy()
    assert not np.all(np.isfinite(yb[7])) and not np.all(np.isfinite(yb[9])) and int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    keep = np.ones(B, dtype=bool)
    keep[[7, 9]] = False
    assert np.array_equal(yb[keep], y.cpu().numpy()[keep])


def test_legacy_family_switch(monkeypatch):
    """RAYEN_SPLIT_BF16 = 0 / 1 / 2, the switch of ABI v2: exact-fp32 kernels only / bf16 triples where the
    measurement accepts them (never f16 pairs) / bf16 triples unmeasured."""
    raw = workloads.make_raw("c3", seed=21)
    for value, want in (("0", 1), ("1", 2), ("2", 2)):
        monkeypatch.setenv("RAYEN_SPLIT_BF16", value)
        _, layer = _layer(raw, torch.float32)
        assert layer.device_pack(torch.device("cuda", 0))[0].info().mfma_f32 == want
    monkeypatch.delenv("RAYEN_SPLIT_BF16")


def _served_random_set(index, family_code):
    """The ``index``-th random set (seeds 1000, 1001, ...) that the given split-operand family (2 / 3) serves under the
    mode in force."""
    found = -1
    for seed in range(1000, 1200):
        raw = _random_set(seed)
        if len(raw["F"]) or raw["y0"].shape[0] > 64:
            continue
        cs, layer = _layer(raw, torch.float32)
        if layer.device_pack(torch.device("cuda", 0))[0].info().mfma_f32 == family_code:
            found += 1
            if found == index:
                return raw
    raise AssertionError("fewer random sets served by the split-operand kernel than expected")


@pytest.mark.parametrize("family", ["pair", "triple"])
@pytest.mark.parametrize("name", ["c2", "c3", "c5r", "served0", "served1", "served2", "served3", "served4", "served5"])
def test_split_operand_kernels_are_fp32_grade(name, family, monkeypatch):
    """The default fp32 forward rebuilds every fp32 product from three f16 MFMA products (operands carried as pairs
    of f16 pieces, 22 bits) or, where the creation-time measurement rejects that (and under fp32_mode 4), from six
    bf16 products (triples, 24 bits).  Measured against the fp64 kernel on the same inputs, the error of either must be
    that of fp32 arithmetic: no worse than the exact-fp32 MFMA kernel's own error (x2 and a 5e-7 floor for sets where
    both are at the rounding level of the outputs)."""
    code = 3 if family == "pair" else 2
    if family == "triple":
        monkeypatch.setenv("RAYEN_FP32_MODE", "4")           # measured, but never the f16 pairs
    raw = _served_random_set(int(name[6:]), code) if name.startswith("served") else workloads.make_raw(name, seed=13)
    cs, layer_split = _layer(raw, torch.float32)
    info = layer_split.device_pack(torch.device("cuda", 0))[0].info()
    assert info.mfma_f32 == code
    # the creation-time measurement that admitted the pack (probe directions incl. +-rows of W, against fp64)
    measured = info.fp32_check_pair if family == "pair" else info.fp32_check_split
    assert 0.0 <= measured <= max(4e-6, 1.5 * info.fp32_check_exact)
    monkeypatch.setenv("RAYEN_FP32_MODE", "1")
    _, layer_exact = _layer(cs, torch.float32)
    monkeypatch.delenv("RAYEN_FP32_MODE")
    _, layer_truth = _layer(cs, torch.float64)
    gen = torch.Generator().manual_seed(15)
    x = torch.empty(20000, cs.n, 1).uniform_(-1.5, 1.5, generator=gen)
    x[:64] *= 1e-3
    y_split = layer_split(x.cuda()).cpu().double().numpy()[:, :, 0]
    y_exact = layer_exact(x.cuda()).cpu().double().numpy()[:, :, 0]
    y_truth = layer_truth(x.double().cuda()).cpu().numpy()[:, :, 0]
    e_split, e_exact = rel_err_rows(y_split, y_truth), rel_err_rows(y_exact, y_truth)
    assert np.max(e_split) <= max(2.0 * np.max(e_exact), 5e-7), (np.max(e_split), np.max(e_exact))
    assert np.mean(e_split) <= max(2.0 * np.mean(e_exact), 1e-7), (np.mean(e_split), np.mean(e_exact))
    # interior samples: y = y0 + NA_E v -- the same fp32 sum up to the order of the additions (triples), or with v
    # rebuilt from its two f16 pieces (22 bits: 2^-23 of the row's largest component)
    assert np.max(rel_err_rows(y_split[:64], y_exact[:64])) <= (3e-7 if family == "triple" else 6e-7)


def test_ill_conditioned_set_on_every_family(monkeypatch):
    """Fuzz set 971 (70 dimensions, one dense quadratic with a large gradient at the interior point, 10 equalities; the
    form the module hands over carries fp32 rounding noise and is numerically rank 40 of 60).  With the round-1 factor
    of that form (pivoted Cholesky, residual 1e-6 |G|) the split-operand kernels were 2e-5 off here and the creation-time
    measurement turned them down; with the eigen-factor (rayen_tiles.h) every family is fp32-grade on it."""
    raw = _random_set(1971)
    gen = torch.Generator().manual_seed(971)
    x = None
    for mode, served in (("0", (2, 3)), ("1", (1,)), ("2", (2,)), ("3", (3,))):
        monkeypatch.setenv("RAYEN_FP32_MODE", mode)
        cs, layer = _layer(raw, torch.float32)
        if x is None:
            x = torch.empty(31, cs.n, 1).uniform_(-2.0, 2.0, generator=gen)
            y_true = _oracle_forward(cs, x.double(), torch.float64)
            y_ref = _oracle_forward(cs, x, torch.float32)
            bound = 2.0 * max(rel_err_rows(y_ref, y_true).max(), 2.5e-6)
        y = layer(x.cuda()).cpu().numpy()[:, :, 0]
        info = layer.device_pack(torch.device("cuda", 0))[0].info()
        assert info.mfma_f32 in served, (mode, info.mfma_f32)
        assert rel_err_rows(y, y_true).max() <= bound, (mode, rel_err_rows(y, y_true).max(), bound)
        if mode == "0":     # measured, and admitted by the rule
            assert 0.0 <= info.fp32_check_pair and 0.0 <= info.fp32_check_split and 0.0 <= info.fp32_check_exact
            mine = info.fp32_check_pair if info.mfma_f32 == 3 else info.fp32_check_split
            assert mine <= max(4e-6, 1.5 * info.fp32_check_exact)
    monkeypatch.delenv("RAYEN_FP32_MODE")


def test_creation_time_measurement_turns_a_family_down():
    """The fallback chain of fp32_mode 0, on a pack made for it: linear rows of size 1e35 are beyond what one global
    power-of-two scale can bring into f16 range (the image overflows), so the f16-pair kernel returns NaN on the
    probes and is turned down; bf16 has fp32's exponent range, so the triples serve the pack -- and agree with fp64."""
    from rayen_amd import pack as _pack
    cs, layer = _layer(workloads.make_raw("c2", seed=5), torch.float32)
    consts = layer.packed_constants()
    lin = [s for s in consts.segments if s.type == 0][0]
    consts.W[lin.row0:lin.row0 + lin.nrows] *= 1e35 / np.abs(consts.W[lin.row0:lin.row0 + lin.nrows]).max()
    dp = _pack.DevicePack(consts, 0)
    info = dp.info()
    assert info.mfma_f32 == 2, info.mfma_f32
    assert not (info.fp32_check_pair <= max(4e-6, 1.5 * info.fp32_check_exact))        # NaN or far off
    assert info.fp32_check_split <= max(4e-6, 1.5 * info.fp32_check_exact)
    gen = torch.Generator().manual_seed(3)
    v = torch.empty(3000, cs.n).uniform_(-1.5, 1.5, generator=gen)
    y = ops.project_raw(v.cuda(), dp)[0].cpu().numpy()
    y_true = ops.project_raw(v.double().cuda(), dp)[0].cpu().numpy()
    assert np.all(np.isfinite(y)) and rel_err_rows(y, y_true).max() <= 1e-5
    # pinned without the measurement the pairs do run (kappa overflows to inf there; y ~ y0 hides it on THIS pack)
    forced = _pack.DevicePack(consts, 0, fp32_mode=3)
    assert forced.info().mfma_f32 == 3 and forced.info().fp32_check_pair == -1.0
    kappa_forced = ops.project_raw(v.cuda(), forced)[1].cpu().numpy()
    kappa_true = ops.project_raw(v.double().cuda(), dp)[1].cpu().numpy()
    assert not np.allclose(kappa_forced, kappa_true, rtol=1e-3)
    forced.close()
    dp.close()


# --------------------------------------------------------------------------- closed-form answers
def _run(layer, v):
    return layer(torch.tensor(v, dtype=torch.float32).unsqueeze(2).cuda()).cpu().numpy()[:, :, 0].astype(np.float64)


def test_known_answer_sphere():
    rho = 2.0
    E = np.eye(3) / rho ** 2
    qc = constraints.ConvexQuadraticConstraint(2 * E, np.zeros((3, 1)), np.array([[-1.0]]))
    cs = constraints.ConvexConstraints(qcs=[qc], y0=np.zeros((3, 1)))
    _, layer = _layer(cs)
    v = np.random.default_rng(0).uniform(-5, 5, size=(777, 3))
    want = v * np.minimum(1.0, rho / np.linalg.norm(v, axis=1))[:, None]
    assert np.max(np.abs(_run(layer, v) - want)) < 1e-5


def test_known_answer_cube_and_halfspace():
    cs = workloads.build_constraints(workloads.cube())
    _, layer = _layer(cs)
    v = np.random.default_rng(1).uniform(-5, 5, size=(500, 3))
    want = 0.5 + v / np.maximum(1.0, 2 * np.max(np.abs(v), axis=1))[:, None]
    assert np.max(np.abs(_run(layer, v) - want)) < 1e-5

    a = np.array([[1.0, -2.0, 0.5]])
    y0 = np.array([[0.1], [0.2], [0.3]])
    lc = constraints.LinearConstraint(a, np.array([[2.0]]), None, None)
    cs = constraints.ConvexConstraints(lc=lc, y0=y0, do_preprocessing_linear=False)
    _, layer = _layer(cs)
    kappa = np.maximum(0.0, (v @ a.T)[:, 0] / (2.0 - (a @ y0).item()))
    want = y0.T + v / np.maximum(1.0, kappa)[:, None]
    assert np.max(np.abs(_run(layer, v) - want)) < 1e-5


def test_known_answer_soc_cone():
    """||(y1,y2)|| <= y3 from y0 = (0,0,1): first positive root of ||v12|| t = 1 + v3 t."""
    M = np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 0]])
    soc = constraints.SOCConstraint(M, np.zeros((3, 1)), np.array([[0.0], [0.0], [1.0]]), np.array([[0.0]]))
    y0 = np.array([[0.0], [0.0], [1.0]])
    cs = constraints.ConvexConstraints(socs=[soc], y0=y0)
    _, layer = _layer(cs)
    v = np.random.default_rng(2).uniform(-5, 5, size=(600, 3))
    kappa = np.maximum(0.0, np.linalg.norm(v[:, :2], axis=1) - v[:, 2])   # 1/t*
    want = y0.T + v / np.maximum(1.0, kappa)[:, None]
    assert np.max(np.abs(_run(layer, v) - want)) < 1e-5


def test_known_answer_psd_cone():
    """[[y1,y2],[y2,y3]] >= 0 from y0 = (1,0,1) = I: kappa = -lambda_min([[v1,v2],[v2,v3]])."""
    F = [np.array([[1.0, 0], [0, 0]]), np.array([[0, 1.0], [1.0, 0]]), np.array([[0, 0], [0, 1.0]]),
         np.zeros((2, 2))]
    y0 = np.array([[1.0], [0.0], [1.0]])
    cs = constraints.ConvexConstraints(lmic=constraints.LMIConstraint(F), y0=y0)
    _, layer = _layer(cs)
    v = np.random.default_rng(3).uniform(-5, 5, size=(600, 3))
    lam_min = 0.5 * (v[:, 0] + v[:, 2]) - np.sqrt(0.25 * (v[:, 0] - v[:, 2]) ** 2 + v[:, 1] ** 2)
    kappa = np.maximum(0.0, -lam_min)
    want = y0.T + v / np.maximum(1.0, kappa)[:, None]
    assert np.max(rel_err_rows(_run(layer, v), want)) <= FP32_TOL


# --------------------------------------------------------------------------- edge cases
def test_edge_shapes_and_strides():
    raw = workloads.make_raw("c2", seed=4)
    cs, layer = _layer(raw)
    # empty batch
    y = layer(torch.zeros(0, cs.n, 1, device="cuda"))
    assert y.shape == (0, cs.k, 1)
    gen = torch.Generator().manual_seed(9)
    for B in (1, 63, 64, 65, 257, 1000):                # ragged tails around the wave / block sizes
        x = torch.empty(B, cs.n, 1).uniform_(-1, 1, generator=gen)
        y = layer(x.cuda()).cpu().numpy()[:, :, 0]
        assert np.max(rel_err_rows(y, _oracle_forward(cs, x, torch.float32))) <= FP32_TOL
    # wider input than n: only the first n columns are read (constraint_module.py:469)
    x = torch.empty(300, cs.n + 5, 1).uniform_(-1, 1, generator=gen)
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    assert np.max(rel_err_rows(y, _oracle_forward(cs, x[:, :cs.n], torch.float32))) <= FP32_TOL
    # image-like input is flattened (constraint_module.py:525)
    x4 = x[:, :cs.n].reshape(300, 4, cs.n // 4, 1)
    y4 = layer(x4.cuda()).cpu().numpy()[:, :, 0]
    assert np.array_equal(y4, layer(x[:, :cs.n].cuda()).cpu().numpy()[:, :, 0])


def test_zero_tiny_and_huge_directions():
    raw = workloads.make_raw("c3", seed=6)
    cs, layer = _layer(raw)
    y0 = cs.y0[:, 0]
    x = torch.zeros(4, cs.n, 1)
    x[1] = 1e-6
    x[2] = 1e6
    x[3, 0] = -1e30
    y = layer(x.cuda()).cpu().numpy()[:, :, 0].astype(np.float64)
    assert np.array_equal(y[0], y0.astype(np.float32).astype(np.float64))   # v = 0 -> exactly y0
    assert np.allclose(y[1], y0 + 1e-6, atol=1e-9)                           # interior: unclipped
    assert np.all(np.isfinite(y))
    assert oracle.max_violation(raw, y) <= VIOLATION_TOL


@pytest.mark.parametrize("name,dtype", [("c3", torch.float32), ("c2", torch.float32), ("c4", torch.float32),
                                        ("c5r", torch.float32), ("c3", torch.float64), ("c5r", torch.float64)])
def test_batches_beyond_2_31_elements(name, dtype):
    """Maximum sizes: direction / output / gradient arrays with more than 2^31 elements (64-bit row offsets in
    every kernel family).  Slices straddling the 2^31-element mark and the ragged tail must come out exactly as
    when those rows are projected on their own (the kernels are row-independent and deterministic)."""
    from rayen_amd import ops
    raw = workloads.make_raw(name, seed=2)
    cs, layer = _layer(raw, dtype)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    width = min(cs.n, cs.k)
    B = (1 << 31) // width + 1500 + 37
    free, _ = torch.cuda.mem_get_info()
    need = (4 if dtype == torch.float32 else 8) * B * (2 * cs.n + 3 * cs.k) + (2 << 30)
    if free < need:
        pytest.skip(f"needs {need >> 30} GiB of HBM")
    gen = torch.Generator(device="cuda").manual_seed(3)
    v = torch.empty(B, cs.n, device="cuda", dtype=dtype).uniform_(-1.5, 1.5, generator=gen)
    mark = (1 << 31) // width
    windows = [(0, 700), (mark - 650, mark + 650), (B - 1037, B)]
    y, kappa, active = ops.project_raw(v, dp, want_active=True)
    assert y.shape == (B, cs.k) and min(y.numel(), v.numel()) > (1 << 31)
    if name == "c3" and dtype == torch.float32 and dp.info().mfma_f32 == 3:
        # (round 6: the W-in-LDS schedule addresses rows through 32-bit buffer offsets and cuts a batch beyond 4 GiB into
        # launches over whole groups; the window around the 2^31-element mark straddles such a cut)
        from rayen_amd import _lib
        assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR_WL
    # config 4: the big batch runs lane-per-sample, a window of 700 rows four lanes per sample -- two
    # algorithms for lambda_max, equal to fp32 accuracy but not bit for bit
    exact = name != "c4"

    def same(a, b):
        if exact:
            return torch.equal(a, b)
        a2, b2 = a.reshape(a.shape[0], -1).double().cpu().numpy(), b.reshape(b.shape[0], -1).double().cpu().numpy()
        return float(np.max(rel_err_rows(a2, b2))) <= FP32_TOL

    for lo, hi in windows:
        y_w, kappa_w, active_w = ops.project_raw(v[lo:hi].clone(), dp, want_active=True)
        assert same(y[lo:hi], y_w), (name, lo, hi)
        assert same(kappa[lo:hi], kappa_w) and torch.equal(active[lo:hi], active_w)
    y_plain, _, _ = ops.project_raw(v, dp, want_active=False)
    for lo, hi in windows:
        # the untracked instance may round differently from the tracked one, but not depend on the batch
        y_w, _, _ = ops.project_raw(v[lo:hi].clone(), dp, want_active=False)
        assert same(y_plain[lo:hi], y_w), (name, lo, hi)
    del y_plain
    assert bool(torch.isfinite(y[::4097]).all())
    g = torch.empty(B, cs.k, device="cuda", dtype=dtype).uniform_(-1, 1, generator=gen)
    grad = ops.backward_raw(v, kappa, active, g, dp)
    for lo, hi in windows:
        grad_w = ops.backward_raw(v[lo:hi].clone(), kappa[lo:hi].clone(), active[lo:hi].clone(), g[lo:hi].clone(), dp)
        if dtype == torch.float32 and dp.info().bwd_f32 == 7:
            # (round 6: the big batch runs on the f16-pair dense-form backward, a window of a few hundred rows on the
            # exact-fp32 kernel it replaces -- two fp32-grade evaluations of the same gradient, not the same bits)
            size = grad_w.abs().amax(dim=1).clamp_min(1e-30)
            assert float(((grad[lo:hi] - grad_w).abs().amax(dim=1) / size).max()) <= 2e-5, (name, lo, hi)
        else:
            assert torch.equal(grad[lo:hi], grad_w), (name, lo, hi)
    tail = y[B - 1037:].cpu().numpy().astype(np.float64)
    assert oracle.max_violation(raw, tail) <= _violation_bound(raw, cs, v[B - 1037:], dtype)


def test_nan_input_raises_like_the_reference():
    raw = workloads.make_raw("c2", seed=4)
    cs, layer = _layer(raw)
    x = torch.zeros(8, cs.n, 1)
    x[3, 2] = float("nan")
    with pytest.raises(AssertionError):
        layer(x.cuda())
    layer(torch.zeros(8, cs.n, 1).cuda())          # the flag is cleared after it fired


def test_module_with_mapper_in_a_sequential():
    """readme.md:76-78 usage: the layer owns an nn.Linear mapper and sits at the end of a model."""
    raw = workloads.make_raw("c2", seed=8)
    cs = workloads.build_constraints(raw)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(3, 64), torch.nn.ReLU(),
                                ConstraintModule(cs, input_dim=64, create_map=True)).cuda()
    x = torch.empty(500, 3, 1).uniform_(-1, 1)
    y = model(x.cuda())
    assert y.shape == (500, cs.k, 1)
    q = model[:3](x.cuda()) @ model[3].mapper.weight.T + model[3].mapper.bias
    y_ref = _oracle_forward(cs, q.detach().cpu().unsqueeze(2), torch.float32)
    assert np.max(rel_err_rows(y.detach().cpu().numpy()[:, :, 0], y_ref)) <= FP32_TOL


# --------------------------------------------------------------------------- full-size properties
@pytest.mark.parametrize("name", ["c3", "c4", "c5"])      # (c5r, the stand-in of rounds 1-2 for c5, added 95 s and no coverage)
def test_full_size_properties(name):
    """BASELINE.json batch sizes: feasibility, scale invariance once clipped, linearity inside."""
    B = min(workloads.CONFIGS[name][2], 262144)
    raw = workloads.make_raw(name, seed=0)
    cs, layer = _layer(raw)
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.empty(B, cs.n, 1, device="cuda").uniform_(-1, 1, generator=gen)
    y = layer(x)
    assert y.shape == (B, cs.k, 1)
    assert bool(torch.isfinite(y).all())
    yc = y[:, :, 0].cpu().numpy()
    # (the whole batch's residuals in fp64 on the device -- helpers.residuals_device --, cross-checked against the oracle's
    # numpy forms on a slice: 262 144 rows x 72 quadratics took 100 s of this test on the host)
    from helpers import residuals_device
    res_dev, rel_dev = residuals_device(raw, y[:, :, 0])
    res = {key: val.cpu().numpy() for key, val in res_dev.items()}
    res_host = oracle.residuals(raw, yc[:2048])
    for key in res_host:
        assert np.allclose(res[key][:2048], res_host[key], rtol=1e-9, atol=1e-12), key
    assert abs(float(rel_dev[:2048].max()) - _relative_violation(raw, yc[:2048])) <= 1e-12
    worst = max(float(np.max(r)) for r in res.values())
    # (residuals are unnormalised and c5's quadratics have |P| ~ 1e2: the yardstick is the violation of the
    # reference's own fp32 output on a slice of the same inputs)
    tol_v = _violation_bound(raw, cs, x[:8192, :, 0], torch.float32)
    head = max(float(np.max(r[:8192])) for r in res.values())          # (the yardstick's own rows)
    assert head <= tol_v
    # ... and the worst rows of the WHOLE batch against the yardstick measured on exactly those rows (round 5: this
    # was `worst <= 2 tol_v` with the slice's yardstick)
    per_row = np.max(np.stack([r.reshape(B, -1).max(axis=1) for r in res.values()]), axis=0)
    top = np.argsort(per_row)[-256:]
    assert int(np.count_nonzero(per_row > tol_v)) <= len(top)
    tol_top = _violation_bound(raw, cs, x[torch.as_tensor(top, device=x.device), :, 0], torch.float32)
    assert worst <= tol_top and float(rel_dev.max()) <= 1e-6
    # clipped samples: y(t v) == y(v) for t > 1 (same ray, same boundary point)
    kappa = layer.computeKappa(x)[:, 0, 0]
    clipped = kappa > 1.5
    # (the corridor set is roomy against U(-1, 1) directions: a quarter of them are clipped, not most)
    assert int(clipped.sum()) > (B // 8 if name == "c5" else B // 2)
    y3 = layer(3.0 * x)[:, :, 0]
    d = (y3 - y[:, :, 0])[clipped].abs().max().item()
    assert d <= 2 * FP32_TOL * max(1.0, float(np.max(np.abs(yc))))    # (two results, each within the parity bar)
    # interior samples: the map is the affine lift y0 + NA_E v
    small = 1e-3 * x
    lift = layer.gety0()[:, 0][None, :] + small[:, :, 0] @ layer.NA_E.T
    # (per row, in fp32 ulps of that row's size: four of them -- the kernel's rounding and the lift's own)
    row_tol = 4.0 * 2.0 ** -23 * lift.abs().amax(dim=1).clamp_min(1.0)
    assert bool(((layer(small)[:, :, 0] - lift).abs().amax(dim=1) <= row_tol).all())
    # order independence: a permuted batch gives the permuted result bit-for-bit
    perm = torch.randperm(B, device="cuda", generator=gen)
    assert torch.equal(layer(x[perm]), y[perm])


def test_hip_graph_capture_replays_the_projection():
    """The op allocates nothing inside the C call and launches on the current stream, so a whole
    forward can be captured in a HIP graph (launch-bound small batches: configs 1-2)."""
    raw = workloads.make_raw("c2", seed=12)
    cs, layer = _layer(raw)
    layer.check_nan = False
    static_x = torch.zeros(4096, cs.n, 1, device="cuda")
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream), torch.no_grad():
        for _ in range(3):
            layer(static_x)
    torch.cuda.current_stream().wait_stream(stream)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        static_y = layer(static_x)
    gen = torch.Generator().manual_seed(4)
    for _ in range(3):
        x = torch.empty(4096, cs.n, 1).uniform_(-1, 1, generator=gen)
        static_x.copy_(x.cuda())
        graph.replay()
        torch.cuda.synchronize()
        y_ref = _oracle_forward(cs, x, torch.float32)
        assert np.max(rel_err_rows(static_y.cpu().numpy()[:, :, 0], y_ref)) <= FP32_TOL


@pytest.mark.parametrize("name,B", [("c2", 4096), ("c3", 8192), ("c5", 8192), ("c5r", 8192)])
def test_bf16_triple_mode(name, B, monkeypatch):
    """fp32_mode 4: bf16 operand triples (6 partial products, fp32 accumulate) where measured fit: same parity bar."""
    monkeypatch.setenv("RAYEN_FP32_MODE", "4")          # read by rayen_pack_create
    raw = workloads.make_raw(name, seed=51)
    cs, layer = _layer(raw)
    gen = torch.Generator().manual_seed(8)
    x = torch.empty(B, cs.n, 1).uniform_(-1, 1, generator=gen)
    x[:4] *= 1e-4
    x[4] = 0
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref = _oracle_forward(cs, x, torch.float32)
    assert np.max(rel_err_rows(y, y_ref)) <= FP32_TOL
    assert oracle.max_violation(raw, y) <= max(VIOLATION_TOL, 3 * oracle.max_violation(raw, y_ref))
    assert np.allclose(y[4], cs.y0[:, 0], atol=1e-6)      # v = 0 -> y0


@pytest.mark.parametrize("name,B", [("c2", 4096), ("c3", 4096), ("c5", 4096), ("c5r", 4096), ("k100_n70", 2000)])
def test_fp64_mfma_and_generic_paths_agree(name, B):
    """fp64: the MFMA kernel (v_mfma_f64_16x16x4_f64) and the generic kernel against the fp64 oracle."""
    raw = _wide_cases()[name] if name.startswith("k") else workloads.make_raw(name, seed=61)
    cs, layer = _layer(raw, torch.float64)
    gen = torch.Generator().manual_seed(10)
    x = torch.empty(B, cs.n, 1, dtype=torch.float64).uniform_(-1, 1, generator=gen)
    x[:3] *= 1e-4
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    y_ref = _oracle_forward(cs, x, torch.float64)
    xf = x.reshape(B, -1).cuda()
    y_auto, kap_auto, act_auto = ops.project_raw(xf, dp)
    y_gen, kap_gen, _ = ops.project_raw(xf, dp, force_generic=True)
    assert np.max(rel_err_rows(y_auto.cpu().numpy(), y_ref)) <= FP64_TOL
    assert np.max(rel_err_rows(y_gen.cpu().numpy(), y_ref)) <= FP64_TOL
    assert torch.allclose(kap_auto, kap_gen, rtol=1e-10, atol=1e-12)
    assert oracle.max_violation(raw, y_auto.cpu().numpy()) <= 1e-11
    if cs.n <= 64:
        assert dp.info().mfma_f64 == 1


@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.float64, FP64_TOL)])
def test_large_subspace_dimension(dtype, tol):
    """n = 800: beyond the LDS tile and the MFMA register budget -> directions read from global memory."""
    raw = workloads.random_lin_quad_soc(k=800, m=120, n_quad=2, n_soc=1, r_M=50, seed=71)
    cs, layer = _layer(raw, dtype)
    gen = torch.Generator().manual_seed(12)
    x = torch.empty(300, cs.n, 1, dtype=torch.float32).uniform_(-1, 1, generator=gen).to(dtype)
    x[0] = 0
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref = _oracle_forward(cs, x, dtype)
    assert np.max(rel_err_rows(y, y_ref)) <= tol
    floor = VIOLATION_TOL if dtype == torch.float32 else 1e-10
    assert oracle.max_violation(raw, y) <= max(floor, 3 * oracle.max_violation(raw, y_ref))


@pytest.mark.parametrize("name", ["c5"])
def test_config5_full_two_million_batch(name):
    """BASELINE.json config 5 (the corridor set; ``c5r`` = the random stand-in of rounds 1-2) at its full size on ONE
    device (the 8-GPU run shards exactly this batch)."""
    B = workloads.CONFIGS[name][2]
    assert B == 2097152
    raw = workloads.make_raw(name, seed=0)
    cs, layer = _layer(raw)
    layer.check_nan = False
    gen = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty(B, cs.n, 1, device="cuda").uniform_(-1, 1, generator=gen)
    y = layer(x)
    assert y.shape == (B, cs.k, 1)
    assert bool(torch.isfinite(y).all())
    sub = y[::64, :, 0].cpu().numpy()                      # residuals of 32768 evenly spaced samples
    # (absolute residuals against the yardstick -- three times what the reference's own fp32 output leaves -- ON THE SAME
    # ROWS: a maximum over more rows is a larger number; the spread sample is held to the relative form)
    head = y[:8192, :, 0].cpu().numpy()
    assert oracle.max_violation(raw, head) <= _violation_bound(raw, cs, x[:8192, :, 0], torch.float32)
    assert _relative_violation(raw, sub) <= 1e-6
    # the same rows in a small batch give the same bits (no dependence on the launch geometry)
    y_small = layer(x[:4096])
    assert torch.equal(y_small, y[:4096])
    from rayen_amd.dist import shard_bounds
    lo, hi = shard_bounds(B, 8, 3)                         # rank 3's shard of the 8-GPU run
    assert torch.equal(layer(x[lo:hi]), y[lo:hi])


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("tag,dtype,tol", [("32", torch.float32, FP32_TOL), ("64", torch.float64, FP64_TOL)])
def test_golden_rayen_old_head(name, tag, dtype, tol):
    """method='RAYEN_old' (constraint_module.py:460-466) against the reference's own outputs."""
    raw, csd, z = load_golden(name)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, method="RAYEN_old", create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    assert layer.getDimAfterMap() == cs.n + 1
    x = torch.cat((_to_my_basis(cs, csd, z["x"], dtype), torch.tensor(z["beta"]).to(dtype)), dim=1)
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    assert np.max(rel_err_rows(y, z["y_old" + tag])) <= tol
    assert oracle.max_violation(raw, y) <= max(VIOLATION_TOL if tag == "32" else 1e-11,
                                               3 * oracle.max_violation(raw, z["y_old" + tag]))


# --------------------------------------------------------------------------- LMI sets of the quad kernel's family
def _lmi_cases():
    rng = np.random.default_rng(12)

    def lmi(k, r):
        F = []
        for _ in range(k):
            T = rng.uniform(-1, 1, size=(r, r))
            F.append((T + T.T) / 2)
        T = rng.uniform(-1, 1, size=(r, r))
        F.append(T @ T.T + 0.5 * np.eye(r))
        return F

    cases = {}
    for name, k, r, m, n_eq in (("r3_lin", 5, 3, 12, 0), ("r7", 6, 7, 0, 0), ("r12_lin_eq", 9, 12, 20, 3),
                                ("r16", 8, 16, 0, 0), ("r24_lin", 6, 24, 9, 0), ("r30_eq", 7, 30, 0, 2)):
        raw = workloads.random_lmi(k, r, seed=0)
        raw["F"] = lmi(k, r)
        if m:
            raw["A1"] = rng.uniform(-1, 1, size=(m, k))
            raw["b1"] = rng.uniform(0.1, 1.0, size=(m, 1))
        if n_eq:
            raw["A2"] = rng.uniform(-1, 1, size=(n_eq, k))
            raw["b2"] = np.zeros((n_eq, 1))                     # y0 = 0 satisfies them
        cases[name] = raw
    return cases


@pytest.mark.parametrize("name", ["r3_lin", "r7", "r12_lin_eq", "r16", "r24_lin", "r30_eq"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, FP32_TOL), (torch.float64, FP64_TOL)])
def test_lmi_with_linear_rows_and_equalities(name, dtype, tol):
    """[linear rows] + one LMI, with and without equality constraints, every size class of the four-lanes-per-
    sample kernel (fp64 above 24 x 24 falls to the lane-per-sample kernel); small and ragged batches."""
    from rayen_amd._lib import RayenError
    raw = _lmi_cases()[name]
    cs, layer = _layer(raw, dtype)
    gen = torch.Generator().manual_seed(6)
    # (30 x 30 in fp64 fits neither the quad kernel's registers nor a lane's LDS column: the wave-per-sample kernel of
    # rayen_lmi_wave.h serves it since round 3 -- tests/test_gpu_lmi_wave.py)
    for B in (1, 67, 1500):
        x = torch.empty(B, cs.n, 1, dtype=torch.float32).uniform_(-2.0, 2.0, generator=gen).to(dtype)
        x[: min(B, 2)] *= 1e-3
        y = layer(x.cuda()).cpu().numpy()[:, :, 0]
        if dtype == torch.float64:
            assert np.max(rel_err_rows(y, _oracle_forward(cs, x, dtype))) <= tol
        else:
            # against the fp64 truth: 1e-5, or twice what LAPACK's fp32 eigvalsh (the reference's arithmetic) leaves
            y_true = _oracle_forward(cs, x.double(), torch.float64)
            bound = _fp32_bound(cs, x, y_true)
            assert rel_err_rows(y, y_true).max() <= bound, (name, B, rel_err_rows(y, y_true).max(), bound)
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        try:
            y_gen, _, _ = ops.project_raw(x[:, :, 0].cuda(), dp, force_generic=True)
        except RayenError:                                      # fp64 beyond ~21 x 21 has no lane-per-sample kernel
            assert dtype == torch.float64 and cs.lmic.all_F[0].shape[0] > 20
            continue
        if dtype == torch.float32:
            assert rel_err_rows(y_gen.cpu().numpy(), y_true).max() <= bound
        else:
            assert np.max(rel_err_rows(y, y_gen.cpu().numpy())) <= 1e-9


# --------------------------------------------------------------------------- randomised constraint sets
def _random_set(seed):
    """A random mix of families, sizes, ranks and equality constraints around a random interior point."""
    rng = np.random.default_rng(seed)
    k = int(rng.choice([3, 5, 8, 13, 16, 24, 31, 32, 33, 40, 48, 64, 70]))
    n_eq = int(rng.integers(0, max(1, k // 4) + 1)) if rng.random() < 0.5 else 0
    raw = workloads._empty(k)
    y0 = rng.uniform(-1, 1, size=(k, 1))
    raw["y0"] = y0
    if n_eq:
        raw["A2"] = rng.uniform(-1, 1, size=(n_eq, k))
        raw["b2"] = raw["A2"] @ y0
    m = int(rng.choice([0, 1, 7, 32, 45, 130]))
    if m:
        raw["A1"] = rng.uniform(-1, 1, size=(m, k))
        raw["b1"] = raw["A1"] @ y0 + rng.uniform(0.1, 1.0, size=(m, 1))
    for _ in range(int(rng.choice([0, 1, 2, 5, 11, 40]))):
        rank = int(rng.choice([1, 2, 3, 4, 5, 8, 9, max(1, k // 2), k]))
        C = rng.uniform(-1, 1, size=(min(rank, k), k))
        P = C.T @ C
        q = rng.uniform(-1, 1, size=(k, 1))
        g0 = 0.5 * y0.T @ P @ y0 + q.T @ y0
        raw["P"].append(P)
        raw["q"].append(q)
        raw["r"].append(-g0 - rng.uniform(0.1, 1.0, size=(1, 1)))
    r_M = int(rng.choice([1, 3, k, k + 5]))                     # (like the reference, one block shape per set)
    for _ in range(int(rng.choice([0, 0, 1, 3]))):
        M = rng.uniform(-1, 1, size=(r_M, k))
        s = rng.uniform(-1, 1, size=(r_M, 1))
        c = rng.uniform(-1, 1, size=(k, 1))
        d = np.linalg.norm(M @ y0 + s) - c.T @ y0 + rng.uniform(0.2, 1.0, size=(1, 1))
        raw["M"].append(M); raw["s"].append(s); raw["c"].append(c); raw["d"].append(d)
    if rng.random() < 0.3 and k <= 16:
        r = int(rng.choice([2, 3, 6, 11]))
        F = []
        for _ in range(k):
            T = rng.uniform(-1, 1, size=(r, r))
            F.append((T + T.T) / 2)
        T = rng.uniform(-1, 1, size=(r, r))
        H = T @ T.T + 0.5 * np.eye(r)
        F.append(H - sum(y0[i, 0] * F[i] for i in range(k)))     # sum y0_i F_i + F_k = H > 0
        raw["F"] = F
    if m == 0 and not raw["P"] and not raw["M"] and not len(raw["F"]):
        raw["A1"] = rng.uniform(-1, 1, size=(4, k))
        raw["b1"] = raw["A1"] @ y0 + rng.uniform(0.1, 1.0, size=(4, 1))
    return raw


def _relative_violation(raw, y):
    """max over constraints and samples of residual / (sum of |terms|)."""
    y = np.asarray(y, dtype=np.float64)
    ay = np.abs(y)
    worst = 0.0
    if raw["A1"] is not None:
        res = y @ raw["A1"].T - raw["b1"].T
        worst = max(worst, float(np.max(res / (ay @ np.abs(raw["A1"]).T + np.abs(raw["b1"]).T))))
    if raw["A2"] is not None:
        res = np.abs(y @ raw["A2"].T - raw["b2"].T)
        worst = max(worst, float(np.max(res / (ay @ np.abs(raw["A2"]).T + np.abs(raw["b2"]).T + 1e-300))))
    for P, q, r in zip(raw["P"], raw["q"], raw["r"]):
        res = 0.5 * np.einsum("bi,ij,bj->b", y, P, y) + y @ q[:, 0] + r[0, 0]
        mag = 0.5 * np.einsum("bi,ij,bj->b", ay, np.abs(P), ay) + ay @ np.abs(q[:, 0]) + abs(r[0, 0])
        worst = max(worst, float(np.max(res / mag)))
    for M, s_, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"]):
        res = np.linalg.norm(y @ M.T + s_.T, axis=1) - y @ c[:, 0] - d[0, 0]
        mag = np.linalg.norm(ay @ np.abs(M).T + np.abs(s_).T, axis=1) + ay @ np.abs(c[:, 0]) + abs(d[0, 0])
        worst = max(worst, float(np.max(res / mag)))
    if len(raw["F"]):
        F = raw["F"]
        S = np.einsum("bi,ijk->bjk", y, np.array(F[:-1])) + F[-1]
        lam = np.linalg.eigvalsh(S)[:, 0]
        mag = np.einsum("bi,i->b", ay, np.array([np.linalg.norm(Fi, 2) for Fi in F[:-1]])) + np.linalg.norm(F[-1], 2)
        worst = max(worst, float(np.max(-lam / mag)))
    return worst


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("RAYEN_FUZZ_SEEDS", "100")))))
def test_random_constraint_sets(seed):
    """Every kernel that serves the set agrees with the oracle (fp32 1e-5, fp64 1e-9), the outputs are feasible,
    and the tracked forward + backward agree with the lane-per-sample backward."""
    raw = _random_set(1000 + seed)
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 31, 64, 65, 1000, 4099]))
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.float64, FP64_TOL)):
        cs, layer = _layer(raw, dtype)
        gen = torch.Generator().manual_seed(seed)
        x = torch.empty(B, cs.n, 1, dtype=torch.float32).uniform_(-2.0, 2.0, generator=gen).to(dtype)
        y = layer(x.cuda()).cpu().numpy()[:, :, 0]
        assert np.all(np.isfinite(y))
        try:
            y_ref = _oracle_forward(cs, x, dtype)
        except AssertionError:
            # the reference op sequence asserts (NaN) when a ray never meets a cone (negative discriminant,
            # CM:342); the kernels give that cone kappa = 0, its exact value: only feasibility can be checked
            assert _relative_violation(raw, y) <= (1e-5 if dtype == torch.float32 else 1e-13), (seed, dtype)
            continue
        if dtype == torch.float32:
            # random sets can be ill-conditioned in fp32 (cancellation in a radicand): judged against the fp64
            # truth under _fp32_bound's yardsticks
            import packed_eval
            y_true = _oracle_forward(cs, x.double(), torch.float64)
            y_const, _, _ = packed_eval.evaluate(layer.packed_constants(), x[:, :, 0].double().numpy())
            bound = _fp32_bound(cs, x, y_true, layer)
            ours = rel_err_rows(y, y_true).max()
            assert ours <= bound, (seed, ours, bound)
        else:
            assert np.max(rel_err_rows(y, y_ref)) <= tol, (seed, dtype)
        # feasibility relative to each constraint's own magnitude (sum of the absolute values of its terms): random
        # sets have |P| up to ~30 k, where one fp32 ulp of y already moves the residual by 1e-3
        if dtype == torch.float32:
            rel_bound = max(5e-6, 4 * _relative_violation(raw, y_ref), 8 * _relative_violation(raw, y_const))
            assert _relative_violation(raw, y) <= min(rel_bound, 1e-4), (seed, dtype)
        else:
            assert _relative_violation(raw, y) <= 1e-13, (seed, dtype)
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        v = x[:, :, 0].cuda()
        y_gen, _, _ = ops.project_raw(v, dp, force_generic=True)
        if dtype == torch.float32:
            assert rel_err_rows(y_gen.cpu().numpy(), y_true).max() <= bound
        else:
            assert np.max(rel_err_rows(y, y_gen.cpu().numpy())) <= 10 * tol
        _, kappa, active = ops.project_raw(v, dp, want_active=True)
        g = torch.empty(B, cs.k, dtype=dtype).uniform_(-1, 1, generator=gen).cuda()
        got = ops.backward_raw(v, kappa, active, g, dp).cpu().double()
        want = ops.backward_raw(v, kappa, active, g, dp, force_generic=True).cpu().double()
        scale = want.abs().amax(1).clamp_min(1e-12)
        err = (got - want).abs().amax(1) / scale
        # (same (kappa, active) record in, so both kernels take the same branch at every sample: no kink exemption)
        assert float(err.max()) <= (2e-4 if dtype == torch.float32 else 1e-9), (seed, dtype, err.max())


@pytest.mark.parametrize("seed", list(range(int(__import__("os").environ.get("RAYEN_FUZZ_SEEDS", "100")) // 2)))
def test_random_modules(seed):
    """Random set x head (RAYEN / RAYEN_old) x mapper (none / nn.Linear of a random width, fused when it can be) x
    batch: the module against the oracle fed with the same mapper output, in fp32 and fp64."""
    raw = _random_set(5000 + seed)
    rng = np.random.default_rng(7000 + seed)
    method = "RAYEN_old" if rng.random() < 0.35 else "RAYEN"
    input_dim = int(rng.choice([0, 0, 4, 8, 20, 33, 64, 100]))
    B = int(rng.choice([1, 33, 64, 127, 2000]))
    for dtype, tol in ((torch.float32, FP32_TOL), (torch.float64, FP64_TOL)):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            cs = workloads.build_constraints(raw)
            torch.manual_seed(seed)
            layer = ConstraintModule(cs, input_dim=input_dim or None, method=method, create_map=bool(input_dim)).cuda()
        finally:
            torch.set_default_dtype(prev)
        gen = torch.Generator().manual_seed(seed)
        width = input_dim or layer.getDimAfterMap()
        x = torch.empty(B, width, dtype=torch.float32).uniform_(-1.5, 1.5, generator=gen).to(dtype)
        with torch.no_grad():
            y = layer(x.cuda()).cpu().numpy()[:, :, 0]
            q = layer.mapper(x.cuda()).cpu() if input_dim else x          # what the projection was fed with
        assert y.shape == (B, cs.k) and np.all(np.isfinite(y))
        buf = oracle.precompute(csd_from_cs(cs), torch.float64)
        try:
            y_true = oracle.forward(buf, q.double().unsqueeze(2), method=method).numpy()[:, :, 0]
        except AssertionError:                                         # a ray that never meets a cone (see above)
            continue
        err = rel_err_rows(y, y_true).max()
        if dtype == torch.float64:
            assert err <= 1e-8, (seed, method, input_dim, err)
        else:
            bound = _fp32_bound(cs, q.unsqueeze(2), y_true, layer, method=method)
            assert err <= bound, (seed, method, input_dim, err, bound)


# --------------------------------------------------------------------------- config 5 with the generator's structure
@pytest.mark.parametrize("B", [8192])
def test_corridor_set_against_truth(B, capsys):
    """BASELINE.json's config 5 restated from the reference's corridor generator (workloads.corridor_spline: clamped
    cubic B-spline, hull regions, 15 boundary equalities, rank-3 velocity / acceleration / jerk limits).  The
    reference's own op sequence is fragile on it -- ``sqrt(rho' delta rho)`` cancels to a slightly negative radicand
    where a control point barely moves: NaN on about half of the directions at fp32 and on some at fp64 (its asserts,
    CM:342-381, stop it first).  So the truth here is the fp64 evaluation of the packed form (tests/packed_eval.py,
    pinned to the reference's fp64 outputs on every golden fixture, config_c5s among them); wherever the oracle IS
    finite at fp64 it must agree with that truth to 1e-9.  The HIP path is held to the north_star's 1e-5 (fp32) and to
    1e-9 (fp64) on EVERY row, and must be finite where the reference is not."""
    import packed_eval
    raw = workloads.make_raw("c5", seed=0)
    cs, layer32 = _layer(raw, torch.float32)
    _, layer64 = _layer(cs, torch.float64)
    gen = torch.Generator().manual_seed(19)
    x = torch.empty(B, cs.n, 1).uniform_(-1.0, 1.0, generator=gen)
    y_true, _, _ = packed_eval.evaluate(layer64.packed_constants(), x[:, :, 0].double().numpy())
    o64 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x.double(), check_nan=False).numpy()[:, :, 0]
    o32 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x, check_nan=False).numpy()[:, :, 0]
    fin64, fin32 = np.isfinite(o64).all(axis=1), np.isfinite(o32).all(axis=1)
    assert fin64.sum() > 0.5 * B
    assert np.max(rel_err_rows(o64[fin64], y_true[fin64])) <= 1e-9          # the truth IS the reference where it is finite
    y32 = layer32(x.cuda()).cpu().numpy()[:, :, 0]
    y64 = layer64(x.double().cuda()).cpu().numpy()[:, :, 0]
    assert np.isfinite(y32).all() and np.isfinite(y64).all()
    e32, e64 = rel_err_rows(y32, y_true), rel_err_rows(y64, y_true)
    info = layer32.device_pack(torch.device("cuda", 0))[0].info()
    with capsys.disabled():
        print(f"\n  [corridor set] reference NaN rows: fp32 {int((~fin32).sum())} / {B}, fp64 {int((~fin64).sum())} / {B}; "
              f"ours: fp32 {e32.max():.2e} (family {info.mfma_f32}), fp64 {e64.max():.2e}")
    assert e64.max() <= FP64_TOL
    assert e32.max() <= FP32_TOL
    assert _relative_violation(raw, y32) <= 1e-6 and _relative_violation(raw, y64) <= 1e-12
    # the f16-pair family serves it (every quadratic carries its own power of two since round 3)
    assert info.mfma_f32 == 3, (info.mfma_f32, info.fp32_check_pair, info.fp32_check_split, info.fp32_check_exact)


def test_zz_fp32_bars_on_record():
    """Last test of the file: how often a yardstick lifted the fp32 bar above the north_star's 1e-5 in this run, and
    by how much; written next to the other GPU artefacts."""
    import json
    import os
    bars = [b for b in BOUND_LOG if b is not None]
    lifted = [b for b in bars if b > FP32_TOL]
    summary = {"bars_applied": len(bars), "lifted_above_1e-5": len(lifted),
               "largest_bar": max(bars) if bars else None, "baseline_configs": RATIO_LOG,
               "lifted": sorted(({"test": w, "bar": b} for w, b in BOUND_WHO if b > FP32_TOL), key=lambda d: -d["bar"])}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "fp32_bound_log.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("\n  [fp32 bars]", json.dumps(summary))
    assert all(b >= FP32_TOL for b in bars)
