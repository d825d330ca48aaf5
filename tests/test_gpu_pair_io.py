"""The f16-pair forward has two schedules of the same arithmetic: rayen_mfma_pair.hip, and the rows of v and y trickled
through LDS under the tile walk (rayen_mfma_pair_io.hip).  Wherever the latter serves a call its outputs must equal the
plain pair kernel's BIT FOR BIT (y, kappa, arg-max record), for every round structure a wave can
see -- one group, first + last, first + middle + last, a ragged last group requested in one burst -- and both must
meet the reference's bar against the oracle (rayen/constraint_module.py:351-474).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from helpers import rel_err_rows, csd_from_cs
from oracle import rayen_oracle as oracle
from rayen_amd import _lib, ops, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _pack(raw):
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, method="RAYEN", create_map=False).to("cuda")
    dp, _ = layer.device_pack(torch.device("cuda", torch.cuda.current_device()))
    return cs, layer, dp


def _sets():
    return {
        "c3": workloads.make_raw("c3", seed=7),                                                # n = 64: 16 blocks per group
        "n32": workloads.random_lin_quad_soc(k=32, m=300, n_quad=3, n_soc=2, seed=17),         # n = 32: 8 blocks per group
        "n32_many_aux": workloads.random_lin_quad_soc(k=32, m=200, n_quad=12, n_soc=3, seed=18),  # 18 aux rows (full patch)
    }


def _misaligned_copy(v):
    """The same rows at an address that is not a multiple of 16 bytes: the trickled kernel declines, the plain one serves."""
    B, n = v.shape
    buf = torch.empty(B * n + 4, dtype=v.dtype, device=v.device)
    w = buf[1:1 + B * n].view(B, n)
    w.copy_(v)
    assert w.data_ptr() % 16 != 0
    return w


def _run(dp, v, want_active):
    y, kappa, active = ops.project_raw(v, dp, want_active=want_active)
    return y, kappa, active, _lib.load().rayen_last_forward_kernel()


@pytest.fixture(autouse=True)
def _schedule_of_rounds_3_to_5():
    """This file is about schedule 1 (the default until round 6: trickled rows / W-stationary for mid-size batches); since
    round 6 the process default is 3 (W in LDS, tests/test_gpu_pair_wl.py), which would take the larger batches here."""
    prev = _lib.load().rayen_pair_schedule(1)
    yield
    _lib.load().rayen_pair_schedule(prev)


SCHEDULES = {"rows": (1, _lib.KERNEL_PAIR_IO)}


@pytest.fixture(params=["rows"])
def schedule(request):
    mode, family = SCHEDULES[request.param]
    prev = _lib.load().rayen_pair_schedule(mode)
    yield family
    _lib.load().rayen_pair_schedule(prev)


# group = 64 rows; a full chip holds 2048 waves: B = 131072 is one group per wave, 262144 two, 393216 three
@pytest.mark.parametrize("B", [64, 1000, 4096 + 37, 32768, 65536 + 101, 131072, 131072 + 64 * 5 + 11, 262144, 393216 + 29, 655360])
@pytest.mark.parametrize("name", ["c3", "n32", "n32_many_aux"])
@pytest.mark.parametrize("want_active", [False, True])
def test_trickled_rows_equal_the_plain_pair_kernel_bit_for_bit(name, B, want_active, schedule):
    if name != "c3" and B > 300000:
        pytest.skip("the round structures are covered on c3")
    cs, layer, dp = _pack(_sets()[name])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    gen = torch.Generator(device="cuda").manual_seed(B)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v[B // 3] = 0.0
    v[B // 2] *= 1e-3
    y1, k1, a1, fam1 = _run(dp, v, want_active)
    y2, k2, a2, fam2 = _run(dp, _misaligned_copy(v), want_active)
    # (batches that do not give every resident wave a 64-row group -- 2048 waves: B < 131072 -- are not for the trickled
    # rows: from two groups per CU on, B >= 32768, the W-stationary schedule takes them where it serves the pack
    # (round 4, third bit-identical schedule), below that the plain kernel)
    if B >= 131072:
        assert fam1 == schedule, "which kernel served the aligned call"
    elif B >= 32768:
        assert fam1 in (_lib.KERNEL_PAIR_WS, _lib.KERNEL_PAIR)
        if name == "c3":
            assert fam1 == _lib.KERNEL_PAIR_WS
    else:
        assert fam1 == _lib.KERNEL_PAIR
    assert fam2 == _lib.KERNEL_PAIR
    assert torch.equal(y1, y2)
    assert torch.equal(k1, k2)
    if want_active:
        assert torch.equal(a1, a2)
    # and against the oracle on a slice (the reference's bar)
    take = torch.cat([torch.arange(0, min(B, 700)), torch.arange(max(B - 700, 0), B)]).unique()
    x = v[take.cuda()].cpu().unsqueeze(2)
    y_ref = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y1[take.cuda()].cpu().numpy(), y_ref)) <= 1e-5


@pytest.mark.parametrize("name", ["c3", "n32"])
def test_trickled_rows_with_padded_leading_dimensions_and_nan_rows(name, schedule):
    """Rows at a stride (ldv, ldy > n, multiples of 4 floats): the LDS-DMA and the stores address every row on its own;
    a NaN row raises the flag and touches no other row."""
    cs, layer, dp = _pack(_sets()[name])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    B, n = 262144 + 77, cs.n
    gen = torch.Generator(device="cuda").manual_seed(3)
    wide = torch.empty(B, n + 8, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v = wide[:, :n]
    out = torch.full((B, n + 12), -7.0, device="cuda")
    y_ref, k_ref, _ = ops.project_raw(_misaligned_copy(v.contiguous()), dp, want_active=False)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR
    y, kappa, _ = ops.project_raw(v, dp, want_active=False, out=out)
    assert _lib.load().rayen_last_forward_kernel() == schedule
    assert torch.equal(out[:, :n], y_ref) and torch.equal(kappa, k_ref)
    assert bool((out[:, n:] == -7.0).all())            # nothing written beyond the k columns
    dp.nan_flag.zero_()
    v2 = v.contiguous().clone()
    v2[B - 5, 3] = float("nan")
    v2[70000, 0] = float("inf")
    y2, _, _ = ops.project_raw(v2, dp, want_active=False)
    assert _lib.load().rayen_last_forward_kernel() == schedule
    assert int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    keep = torch.ones(B, dtype=torch.bool, device="cuda")
    keep[B - 5] = False
    keep[70000] = False
    assert torch.equal(y2[keep], y_ref[keep])


# --------------------------------------------------------------------------------------------------------------
# rows stored back to back, n <= 32, NA_E = I or not (mfma_pair_iof_kernel: flat blocks, whole-line stores)
# --------------------------------------------------------------------------------------------------------------
def _flat_sets():
    eq = workloads.corridor_like(k=28, n_eq=8, m=330, n_quad=10, rank=3, seed=31)          # n = 20 of k = 28, 14 tiles
    return {
        "c5r": workloads.make_raw("c5r", seed=9),                                                # n = 30 of k = 45, 72 packed quadratics
        "c5": workloads.make_raw("c5", seed=0),                                                  # the corridor set: 1050 rows, 47 tiles
        "eq_n20": eq,
        "id_n24": workloads.random_lin_quad_soc(k=24, m=260, n_quad=3, n_soc=1, seed=32),      # NA_E = I, ragged n
        "id_n30_many": workloads.random_lin_quad_soc(k=30, m=300, n_quad=6, n_soc=2, seed=33),
        "c2": workloads.make_raw("c2", seed=10),                                               # n = 16: four chunks per group
    }


@pytest.mark.parametrize("B", [64, 777, 131072, 131072 + 64 * 3 + 5, 262144, 393216 + 17])
@pytest.mark.parametrize("name", ["c5r", "c5", "eq_n20", "id_n24", "id_n30_many", "c2"])
@pytest.mark.parametrize("want_active", [False, True])
def test_flat_rows_equal_the_plain_pair_kernel_bit_for_bit(name, B, want_active):
    if name not in ("c5r", "id_n24") and B > 300000:
        pytest.skip("the round structures are covered on two sets")
    cs, layer, dp = _pack(_flat_sets()[name])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    gen = torch.Generator(device="cuda").manual_seed(B + 1)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v[B // 3] = 0.0
    v[B // 2] *= 1e-3
    assert v.data_ptr() % 16 == 0
    y1, k1, a1, fam1 = _run(dp, v, want_active)
    y2, k2, a2, fam2 = _run(dp, _misaligned_copy(v), want_active)
    assert fam2 == _lib.KERNEL_PAIR
    if fam1 != _lib.KERNEL_PAIR_IO:
        pytest.skip("this shape is not served by the flat-row kernel (too few tiles for its chunks)")
    assert torch.equal(y1, y2)
    assert torch.equal(k1, k2)
    if want_active:
        assert torch.equal(a1, a2)
    # the reference's bar, against the fp64 oracle with the reference's own fp32 error as the yardstick (random sets)
    take = torch.cat([torch.arange(0, min(B, 500)), torch.arange(max(B - 500, 0), B)]).unique()
    x = v[take.cuda()].cpu().unsqueeze(2)
    # (rows on which the reference's op sequence is NaN -- the corridor set, tests/test_gpu_parity.py::
    # test_corridor_set_against_truth -- are left to that test)
    y_true = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x.double(), check_nan=False).numpy()[:, :, 0]
    y_ref = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x, check_nan=False).numpy()[:, :, 0]
    ok = np.isfinite(y_true).all(axis=1) & np.isfinite(y_ref).all(axis=1)
    if ok.sum() < 0.3 * len(ok):
        assert name == "c5"          # (the reference's fp32 op sequence is NaN on 40-100 % of the corridor set's rows, by host BLAS path)
        return
    bound = max(1e-5, 2.0 * float(np.max(rel_err_rows(y_ref[ok], y_true[ok]))))
    assert np.max(rel_err_rows(y1[take.cuda()].cpu().numpy()[ok], y_true[ok])) <= bound


def test_flat_rows_served_sets_are_really_served():
    """Config 5 and the ragged identity set must be on the flat-row kernel at the BASELINE batch (no silent fallback)."""
    for name in ("c5r", "c5", "id_n24", "eq_n20"):
        cs, layer, dp = _pack(_flat_sets()[name])
        v = torch.empty(262144, cs.n, device="cuda").uniform_(-1, 1)
        _, _, _, fam = _run(dp, v, False)
        assert fam == _lib.KERNEL_PAIR_IO, name
        # a NaN row raises the flag and touches no other row
        dp.nan_flag.zero_()
        y_ref, _, _ = ops.project_raw(v, dp, want_active=False)
        v2 = v.clone()
        v2[1234, 2] = float("nan")
        y2, _, _ = ops.project_raw(v2, dp, want_active=False)
        assert int(dp.nan_flag.item()) == 1
        dp.nan_flag.zero_()
        keep = torch.ones(v.shape[0], dtype=torch.bool, device="cuda")
        keep[1234] = False
        assert torch.equal(y2[keep], y_ref[keep])


def test_dynamic_lds_promise_survives_a_later_smaller_pack():
    """hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel instance, not to the pack: a
    pack with smaller rows created AFTER config 5 must not lower what config 5's launches of the same flat-row forward
    instance and of the resident pair backward ask for.  Large pack, small pack, then the large one again."""
    big = workloads.make_raw("c5", seed=0)
    cs_b, layer_b, dp_b = _pack(big)
    v = torch.empty(131072 + 64, cs_b.n, device="cuda").uniform_(-1, 1)
    g = torch.empty(v.shape[0], cs_b.k, device="cuda").uniform_(-1, 1)
    y0, k0, a0, fam0 = _run(dp_b, v, True)
    gv0 = ops.backward_raw(v, k0, a0, g, dp_b)
    small = workloads.corridor_like(k=24, n_eq=4, m=96, n_quad=6, rank=3, seed=4)     # n = 20, same instances, less LDS
    cs_s, layer_s, dp_s = _pack(small)
    vs = torch.empty(131072 + 64, cs_s.n, device="cuda").uniform_(-1, 1)
    ys, ks, as_, fam_s = _run(dp_s, vs, True)
    ops.backward_raw(vs, ks, as_, torch.ones(vs.shape[0], cs_s.k, device="cuda"), dp_s)
    y1, k1, a1, fam1 = _run(dp_b, v, True)
    gv1 = ops.backward_raw(v, k1, a1, g, dp_b)
    torch.cuda.synchronize()
    assert fam0 == fam1 == _lib.KERNEL_PAIR_IO
    assert torch.equal(y0, y1) and torch.equal(k0, k1) and torch.equal(a0, a1) and torch.equal(gv0, gv1)


# --------------------------------------------------------------------------------------------------------------
# the same launch many times (round 4): 22 of 3 000 launches of the flat-row kernel's TRACK instance on config 5 at
# B = 655 360 returned y0 itself in 16 rows of one column (a packed fma of the NA_E write-out: rayen_mfma_pair_io.hip),
# never repeatably -- one launch per shape cannot see that
# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,B,reps", [("c5", 655360, 300), ("c5r", 393216, 150), ("c3", 393216, 100)])
def test_trickled_rows_equal_the_plain_kernel_on_every_one_of_many_launches(name, B, reps):
    raw = _flat_sets()[name] if name != "c3" else _sets()["c3"]
    cs, layer, dp = _pack(raw)
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    gen = torch.Generator(device="cuda").manual_seed(B)
    buf = torch.empty(B * cs.n + 4, device="cuda")
    w = buf[1:1 + B * cs.n].view(B, cs.n)
    assert w.data_ptr() % 16 != 0
    mismatching = 0
    for rep in range(reps):
        v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
        y1, k1, a1, fam1 = _run(dp, v, True)                  # (the arg-max record: the instance that failed)
        assert fam1 == _lib.KERNEL_PAIR_IO
        w.copy_(v)
        y2, k2, a2, fam2 = _run(dp, w, True)
        assert fam2 == _lib.KERNEL_PAIR
        mismatching += int(not (torch.equal(y1, y2) and torch.equal(k1, k2) and torch.equal(a1, a2)))
    assert mismatching == 0, f"{mismatching} of {reps} launches differ from the plain kernel"
