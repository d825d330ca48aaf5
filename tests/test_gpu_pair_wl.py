"""The W-in-LDS schedule of the f16-pair forward (rayen_mfma_pair_wl.hip, round 6: the image of W copied into LDS once per
workgroup, a lane's rows straight from / to memory, groups dealt to the waves of a workgroup on demand).  Wherever it serves
a call its outputs must equal the plain pair kernel's BIT FOR BIT (y, kappa, arg-max record) -- whatever the batch (ragged
last groups, fewer groups than waves, many groups per wave), the leading dimensions, NaN rows -- and meet the reference's
bar against the oracle (rayen/constraint_module.py:351-474).  The order in which the waves of a workgroup claim groups
varies from launch to launch: the results must not.  Needs an MI355X."""
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
        "c3": workloads.make_raw("c3", seed=7),                                                   # n = 64: shared tiles, 8 aux rows
        "n32": workloads.random_lin_quad_soc(k=32, m=300, n_quad=3, n_soc=2, seed=17),            # n = 32
        "n32_many_aux": workloads.random_lin_quad_soc(k=32, m=200, n_quad=12, n_soc=3, seed=18),  # 18 aux rows
        # n = k below the padded width, in whole 16-byte pieces (round 6): the pieces beyond a row's n columns are out of range
        "c2": workloads.make_raw("c2", seed=1),                                                   # n = 16 (BASELINE configs[1])
        "n20": workloads.random_lin_quad_soc(k=20, m=90, n_quad=2, n_soc=1, seed=21),
        "n28": workloads.random_lin_quad_soc(k=28, m=64, n_quad=3, n_soc=0, seed=22),
        "n36": workloads.random_lin_quad_soc(k=36, m=100, n_quad=2, n_soc=1, seed=23),            # NKK = 2, 36 of 64 columns
        "n60": workloads.random_lin_quad_soc(k=60, m=128, n_quad=3, n_soc=2, seed=24),
        "n64_many_aux": workloads.random_lin_quad_soc(k=64, m=32, n_quad=5, n_soc=3, seed=25),     # 11 aux rows at NKK = 2 (sixteen fit)
    }


def _misaligned_copy(v):
    B, n = v.shape
    buf = torch.empty(B * n + 4, dtype=v.dtype, device=v.device)
    w = buf[1:1 + B * n].view(B, n)
    w.copy_(v)
    assert w.data_ptr() % 16 != 0
    return w


def _run(dp, v, want_active):
    y, kappa, active = ops.project_raw(v, dp, want_active=want_active)
    return y, kappa, active, _lib.load().rayen_last_forward_kernel()


@pytest.fixture
def lds_schedule():
    prev = _lib.load().rayen_pair_schedule(3)
    yield _lib.KERNEL_PAIR_WL
    _lib.load().rayen_pair_schedule(prev)


def test_the_w_in_lds_schedule_is_the_default():
    assert _lib.load().rayen_pair_schedule(-1) == 3


# a group = 32 rows; 256 CUs x 16 waves: B = 131072 is one group per wave; the schedule serves every batch size (one workgroup
# per CU as soon as there is a group for it: below 4 096 groups some waves, below 256 some CUs have nothing to do)
@pytest.mark.parametrize("B", [1, 31, 33, 1000, 4096 + 7, 8192, 32768 + 5, 65536, 98304, 98304 + 17, 131072, 131072 + 32 * 5 + 11,
                               262144, 262144 - 1, 393216 + 29, 1048576 + 3])
@pytest.mark.parametrize("name", ["c3", "n32", "n32_many_aux", "c2", "n20", "n28", "n36", "n60", "n64_many_aux"])
@pytest.mark.parametrize("want_active", [False, True])
def test_w_in_lds_equals_the_plain_pair_kernel_bit_for_bit(name, B, want_active, lds_schedule):
    if name != "c3" and B > 300000:
        pytest.skip("the round structures are covered on c3")
    if name in ("n20", "n28", "n36", "n60", "c2", "n64_many_aux") and B not in (1, 33, 4096 + 7, 65536, 131072 + 32 * 5 + 11):
        pytest.skip("the ragged widths are covered on five batch shapes")
    cs, layer, dp = _pack(_sets()[name])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    gen = torch.Generator(device="cuda").manual_seed(B)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v[B // 3] = 0.0
    v[B // 2] *= 1e-3
    v[B - 1] *= 1e4
    y1, k1, a1, fam1 = _run(dp, v, want_active)
    assert fam1 == lds_schedule, "which kernel served the aligned call"
    y2, k2, a2, fam2 = _run(dp, _misaligned_copy(v), want_active)
    assert fam2 == _lib.KERNEL_PAIR
    assert torch.equal(y1, y2)
    assert torch.equal(k1, k2)
    if want_active:
        assert torch.equal(a1, a2)
    take = torch.cat([torch.arange(0, min(B, 700)), torch.arange(max(B - 700, 0), B)]).unique()
    x = v[take.cuda()].cpu().unsqueeze(2)
    y_ref = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y1[take.cuda()].cpu().numpy(), y_ref)) <= 1e-5


def test_small_batches_are_served_too_and_what_it_cannot_address_is_not(lds_schedule):
    """Round 6: no batch threshold any more (a small batch leaves waves idle, it does not lose: DESIGN.md 4.0).  What the
    buffer descriptors cannot address (4 GiB of rows) and rows that are not 16-byte aligned go to the other schedules."""
    cs, layer, dp = _pack(_sets()["c3"])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    for B in (1, 64, 4096, 65536):
        v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5)
        assert _run(dp, v, False)[3] == lds_schedule, B
    v = torch.empty(4096, cs.n, device="cuda").uniform_(-1.5, 1.5)
    assert _run(dp, _misaligned_copy(v), False)[3] == _lib.KERNEL_PAIR


@pytest.mark.parametrize("name", ["c3", "n32", "c2", "n20", "n36"])
def test_w_in_lds_with_padded_leading_dimensions_and_nan_rows(name, lds_schedule):
    """Rows at a stride (ldv, ldy > n, multiples of 4 floats); a NaN / Inf row raises the flag and touches no other row; rows
    beyond the batch are never written."""
    cs, layer, dp = _pack(_sets()[name])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    B, n = 262144 + 77, cs.n
    gen = torch.Generator(device="cuda").manual_seed(3)
    wide = torch.empty(B, n + 8, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v = wide[:, :n]
    out = torch.full((B + 40, n + 12), -7.0, device="cuda")
    y_ref, k_ref, _ = ops.project_raw(_misaligned_copy(v.contiguous()), dp, want_active=False)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR
    y, kappa, _ = ops.project_raw(v, dp, want_active=False, out=out[:B])
    assert _lib.load().rayen_last_forward_kernel() == lds_schedule
    assert torch.equal(out[:B, :n], y_ref) and torch.equal(kappa, k_ref)
    assert bool((out[:, n:] == -7.0).all())            # nothing written beyond the k columns ...
    assert bool((out[B:] == -7.0).all())               # ... nor beyond the batch (the last group is ragged)
    dp.nan_flag.zero_()
    v2 = v.contiguous().clone()
    v2[B - 5, 3] = float("nan")
    v2[70000, 0] = float("inf")
    y2, _, _ = ops.project_raw(v2, dp, want_active=False)
    assert _lib.load().rayen_last_forward_kernel() == lds_schedule
    assert int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    keep = torch.ones(B, dtype=torch.bool, device="cuda")
    keep[B - 5] = False
    keep[70000] = False
    assert torch.equal(y2[keep], y_ref[keep])
    assert bool(torch.isnan(y2[B - 5]).any()) and bool(torch.isnan(y2[70000]).any())


def test_w_in_lds_is_the_same_on_every_launch(lds_schedule):
    """Which wave of a workgroup claims which group differs from launch to launch (a counter in LDS); a group's results do
    not depend on it.  200 launches into fresh buffers, every byte compared."""
    cs, layer, dp = _pack(_sets()["c3"])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    B = 262144 + 4096 + 13
    v = torch.empty(B, cs.n, device="cuda").uniform_(-2.0, 2.0, generator=torch.Generator(device="cuda").manual_seed(11))
    y0, k0, a0, fam = _run(dp, v, True)
    assert fam == lds_schedule
    for _ in range(200):
        out = torch.full((B, cs.k), float("nan"), device="cuda")
        y, k, a = ops.project_raw(v, dp, want_active=True, out=out)
        assert torch.equal(y, y0) and torch.equal(k, k0) and torch.equal(a, a0)


def test_module_forward_takes_the_w_in_lds_schedule_on_the_headline_shape():
    """BASELINE.json's headline call (config 3, B = 262 144, fp32, the module's own forward) runs on this kernel by default and
    meets the fp32 bar against the fp64 oracle on a slice."""
    raw = workloads.make_raw("c3", seed=0)
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, method="RAYEN", create_map=False).to("cuda")
    x = torch.empty(262144, cs.n, 1, device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(5))
    with torch.no_grad():
        y = layer(x)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR_WL
    take = torch.arange(0, 262144, 257)
    y_ref = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x[take.cuda()].double().cpu()).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y[take.cuda(), :, 0].cpu().numpy(), y_ref)) <= 1e-5


@pytest.mark.parametrize("B", [98304 + 1, 131072 + 31, 262144 - 13])
def test_nothing_is_written_or_used_beyond_a_ragged_batch(B, lds_schedule):
    """The rows of the last group beyond the batch are out of range of the buffer descriptors (round 6): their loads return
    zero and their stores are dropped by the hardware.  y and v are views of larger allocations whose tails hold a canary /
    NaN: the canary must survive, the NaN must not be seen (nan_flag stays clear, the outputs equal the plain kernel's)."""
    cs, layer, dp = _pack(_sets()["c3"])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    gen = torch.Generator(device="cuda").manual_seed(B + 5)
    v_all = torch.full((B + 64, cs.n), float("nan"), device="cuda")
    v_all[:B].uniform_(-1.5, 1.5, generator=gen)
    y_all = torch.full((B + 64, cs.k), 777.0, device="cuda")
    v = v_all[:B]
    out = y_all[:B]
    y, kappa, _ = ops.project_raw(v, dp, out=out)
    assert _lib.load().rayen_last_forward_kernel() == lds_schedule
    assert y.data_ptr() == out.data_ptr()
    assert bool((y_all[B:] == 777.0).all()), "rows beyond the batch were written"
    assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(kappa).all())
    y2, k2, _ = ops.project_raw(_misaligned_copy(v), dp)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR
    assert torch.equal(y, y2) and torch.equal(kappa, k2)


@pytest.mark.parametrize("name,in_dim", [("c3", 64), ("c3", 36), ("c3", 20), ("c2", 8), ("n32", 32), ("n36", 12), ("n60", 64)])
def test_mapped_w_in_lds_against_the_plain_mapped_kernel_and_against_itself(name, in_dim):
    """The module's mapper in front of the walk (rayen/constraint_module.py:259-263, 525), round 6: the W-in-LDS schedule with
    the mapper's image next to W's (rayen_mfma_pair_wl.hip, NKX > 0).
      * v = Wm x + b equals the mapped instances of rayen_mfma_pair.hip (rayen_pair_schedule(0)) bit for bit: same image of the
        mapper, same order of products;
      * y, kappa and the arg-max record equal the UNMAPPED W-in-LDS kernel fed with that v bit for bit: v leaves as an exact
        power-of-two multiple of the accumulators the pieces are split from;
      * against the plain mapped kernel y agrees to fp32 rounding only -- that one walks the pack's image WITHOUT shared tiles
        (other row order inside the sums of squares);
      * the instance without the record returns the same y."""
    lib = _lib.load()
    cs = workloads.build_constraints(_sets()[name])
    torch.manual_seed(in_dim)
    layer = ConstraintModule(cs, input_dim=in_dim, create_map=True).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    if dp.info().mfma_f32 != 3 or dp.mapper_mode(in_dim) != 2:
        pytest.skip("the f16-pair family does not fuse this mapper")
    pid = ops.register_pack(dp)
    w, b = layer.mapper.weight.detach(), layer.mapper.bias.detach()
    prev = lib.rayen_pair_schedule(-1)
    try:
        for B in (1, 33, 1000, 65536 + 7, 262144):
            x = torch.empty(B, in_dim, device="cuda").uniform_(-2.0, 2.0, generator=torch.Generator(device="cuda").manual_seed(B))
            x[B // 2] *= 1e-3
            lib.rayen_pair_schedule(3)
            y, kappa, active, v = ops.ray_project_mapped(x, w, b, pid, True)
            assert lib.rayen_last_forward_kernel() == _lib.KERNEL_PAIR_WL, (name, in_dim, B)
            y_plain, _, _, _ = ops.ray_project_mapped(x, w, b, pid, False)
            assert torch.equal(y_plain, y), (name, in_dim, B, "without the record")
            y_u, kappa_u, active_u = ops.project_raw(v, dp, want_active=True)
            assert lib.rayen_last_forward_kernel() == _lib.KERNEL_PAIR_WL
            assert torch.equal(y, y_u) and torch.equal(kappa, kappa_u) and torch.equal(active, active_u), (name, in_dim, B, "unmapped on v")
            lib.rayen_pair_schedule(0)
            y0_, kappa0, active0, v0 = ops.ray_project_mapped(x, w, b, pid, True)      # the mapped instances of rayen_mfma_pair.hip
            assert torch.equal(v, v0), (name, in_dim, B, "v")
            size = y0_.abs().amax(dim=1).clamp_min(1e-30)
            assert float(((y - y0_).abs().amax(dim=1) / size).max()) <= 2e-6, (name, in_dim, B, "y against the plain mapped kernel")
    finally:
        lib.rayen_pair_schedule(prev)
