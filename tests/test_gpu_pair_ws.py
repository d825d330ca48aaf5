"""The W-stationary schedule of the f16-pair forward (rayen_mfma_pair_ws8.hip, schedule 2: the tiles of W resident in the
registers of a workgroup's eight waves, the batch streamed through a shared B-operand image in LDS) issues the MFMAs of
every row tile in rayen_mfma_pair.hip's order on the same operands, so wherever it serves a call its outputs must equal
the plain pair kernel's BIT FOR BIT -- for every structure a workgroup's loop can see (one group per workgroup, two, many, a
ragged last group), padded leading dimensions, NaN rows -- and meet the reference's bar against the oracle
(rayen/constraint_module.py:351-474).  Needs an MI355X."""
import numpy as np
import pytest
import torch

from helpers import rel_err_rows, csd_from_cs
from oracle import rayen_oracle as oracle
from rayen_amd import _lib, ops, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _schedule_of_rounds_3_to_5():
    """This file is about schedule 1 (the default until round 6: trickled rows / W-stationary for mid-size batches); since
    round 6 the process default is 3 (W in LDS, tests/test_gpu_pair_wl.py), which would take the larger batches here."""
    prev = _lib.load().rayen_pair_schedule(1)
    yield
    _lib.load().rayen_pair_schedule(prev)


def _pack(raw):
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, method="RAYEN", create_map=False).to("cuda")
    dp, _ = layer.device_pack(torch.device("cuda", torch.cuda.current_device()))
    return cs, layer, dp


def _sets():
    return {
        "c3": workloads.make_raw("c3", seed=7),                                             # 17 tiles: 5 + 4 + 4 + 4
        "lin_only": workloads.random_lin_quad_soc(k=64, m=300, n_quad=0, n_soc=0, seed=41),  # 10 tiles of rows, no aux tile
        "few": workloads.random_lin_quad_soc(k=64, m=40, n_quad=1, n_soc=1, seed=42),        # 7 tiles: the three-stage instance
        "quads": workloads.random_lin_quad_soc(k=64, m=100, n_quad=6, n_soc=1, seed=43),     # 19 tiles
        "n32": workloads.random_lin_quad_soc(k=32, m=300, n_quad=3, n_soc=2, seed=17),       # n = 32 (eight waves only)
    }


def _misaligned_copy(v):
    """The same rows at an address that is not a multiple of 16 bytes: only the plain pair kernel serves it."""
    B, n = v.shape
    buf = torch.empty(B * n + 4, dtype=v.dtype, device=v.device)
    w = buf[1:1 + B * n].view(B, n)
    w.copy_(v)
    assert w.data_ptr() % 16 != 0
    return w


def _run(dp, v, want_active=False, **kw):
    y, kappa, active = ops.project_raw(v, dp, want_active=want_active, **kw)
    if want_active:
        return y, kappa, active, _lib.load().rayen_last_forward_kernel()
    return y, kappa, _lib.load().rayen_last_forward_kernel()


@pytest.fixture(autouse=True, params=[2], ids=["eight_waves"])
def w_stationary_schedule(request):
    prev = _lib.load().rayen_pair_schedule(request.param)
    yield request.param
    _lib.load().rayen_pair_schedule(prev)


def _served(dp, cs):
    v = torch.zeros(65536, cs.n, device="cuda")
    return _run(dp, v)[2] == _lib.KERNEL_PAIR_WS


# a group = 64 rows, one workgroup per CU (256): B = 32768 is two groups per workgroup (the least the kernel takes),
# 262144 sixteen; ragged batches leave some workgroups a group short and the last group part empty
@pytest.mark.parametrize("B", [32768, 32768 + 64 * 7 + 5, 49152 + 1, 131072, 262144, 262144 + 64 * 100 + 63, 655360 + 17])
@pytest.mark.parametrize("name", ["c3", "lin_only", "few", "quads", "n32"])
def test_w_stationary_equals_the_plain_pair_kernel_bit_for_bit(name, B):
    if name != "c3" and B > 300000:
        pytest.skip("the long loops are covered on c3")
    cs, layer, dp = _pack(_sets()[name])
    if dp.info().mfma_f32 != 3:
        pytest.skip("the f16-pair family does not serve this pack")
    if not _served(dp, cs):
        pytest.skip("the W-stationary kernel does not serve this pack (tiles per wave beyond the compiled instances)")
    gen = torch.Generator(device="cuda").manual_seed(B)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v[B // 3] = 0.0
    v[B // 2] *= 1e-3
    v[B // 5] *= 64.0
    y1, k1, fam1 = _run(dp, v)
    y2, k2, fam2 = _run(dp, _misaligned_copy(v))
    assert fam1 == _lib.KERNEL_PAIR_WS and fam2 == _lib.KERNEL_PAIR
    assert torch.equal(k1, k2)
    assert torch.equal(y1, y2)
    # and against the oracle on a slice (the reference's bar)
    take = torch.cat([torch.arange(0, min(B, 700)), torch.arange(max(B - 700, 0), B)]).unique()
    x = v[take.cuda()].cpu().unsqueeze(2)
    y_ref = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y1[take.cuda()].cpu().numpy(), y_ref)) <= 1e-5


def test_c3_is_served_at_the_baseline_batch(w_stationary_schedule):
    cs, layer, dp = _pack(_sets()["c3"])
    v = torch.empty(262144, cs.n, device="cuda").uniform_(-1, 1)
    assert _run(dp, v)[2] == _lib.KERNEL_PAIR_WS
    # the training forward (arg-max record) too; small batches stay on the plain kernel
    ops.project_raw(v, dp, want_active=True)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR_WS
    assert _run(dp, v[:4096])[2] == _lib.KERNEL_PAIR


@pytest.mark.parametrize("B", [32768 + 64 * 7 + 5, 262144])
@pytest.mark.parametrize("name", ["c3", "lin_only", "few", "quads", "n32", "n32_packed"])
def test_arg_max_record_equals_the_plain_pair_kernel(name, B, w_stationary_schedule):
    """Training forward: kappa and the (segment, row) record, ties included (a cube-like set has them)."""
    sets = dict(_sets())
    sets["n32"] = workloads.random_lin_quad_soc(k=32, m=300, n_quad=3, n_soc=2, seed=17)
    sets["n32_packed"] = workloads.make_raw("c2", seed=10) if False else workloads.random_lin_quad_soc(k=32, m=64, n_quad=6, n_soc=1, seed=19)
    cs, layer, dp = _pack(sets[name])
    if dp.info().mfma_f32 != 3 or not _served(dp, cs):
        pytest.skip("not served by the W-stationary kernel")
    gen = torch.Generator(device="cuda").manual_seed(B + 5)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v[7] = 0.0
    v[11] = 1.0            # ties between rows of symmetric sets
    y1, k1, a1, fam1 = _run(dp, v, want_active=True)
    y2, k2, a2, fam2 = _run(dp, _misaligned_copy(v), want_active=True)
    assert fam1 == _lib.KERNEL_PAIR_WS and fam2 == _lib.KERNEL_PAIR
    assert torch.equal(k1, k2) and torch.equal(y1, y2)
    assert torch.equal(a1, a2)


@pytest.mark.parametrize("name", ["c3", "quads"])
def test_w_stationary_with_padded_leading_dimensions_and_nan_rows(name):
    """Rows at a stride (ldv, ldy > n, multiples of 4 floats); a NaN row raises the flag and touches no other row."""
    cs, layer, dp = _pack(_sets()[name])
    if dp.info().mfma_f32 != 3 or not _served(dp, cs):
        pytest.skip("not served by the W-stationary kernel")
    B, n = 262144 + 77, cs.n
    gen = torch.Generator(device="cuda").manual_seed(3)
    wide = torch.empty(B, n + 8, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    v = wide[:, :n]
    out = torch.full((B, n + 12), -7.0, device="cuda")
    y_ref, k_ref, fam = _run(dp, _misaligned_copy(v.contiguous()))
    assert fam == _lib.KERNEL_PAIR
    y, kappa, fam = _run(dp, v, out=out)
    assert fam == _lib.KERNEL_PAIR_WS
    assert torch.equal(out[:, :n], y_ref) and torch.equal(kappa, k_ref)
    assert bool((out[:, n:] == -7.0).all())            # nothing written beyond the k columns
    dp.nan_flag.zero_()
    v2 = v.contiguous().clone()
    v2[B - 5, 3] = float("nan")
    v2[70000, 0] = float("inf")
    y2, _, fam = _run(dp, v2)
    assert fam == _lib.KERNEL_PAIR_WS
    assert int(dp.nan_flag.item()) == 1
    dp.nan_flag.zero_()
    keep = torch.ones(B, dtype=torch.bool, device="cuda")
    keep[B - 5] = False
    keep[70000] = False
    assert torch.equal(y2[keep], y_ref[keep])


def test_repeated_launches_give_the_same_bits():
    """Counted vmcnt waits behind LDS-DMA and two skewed teams of waves: a race shows as run-to-run differences first."""
    cs, layer, dp = _pack(_sets()["c3"])
    gen = torch.Generator(device="cuda").manual_seed(11)
    v = torch.empty(262144, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    y0, k0, fam = _run(dp, v)
    assert fam == _lib.KERNEL_PAIR_WS
    y0, k0 = y0.clone(), k0.clone()
    for _ in range(20):
        y, k, _ = _run(dp, v)
        assert torch.equal(y, y0) and torch.equal(k, k0)
