"""The fp32 backward on f16 pairs (rayen_amd/csrc/rayen_mfma_bwdp.hip): sets with n <= 32 whose quadratics all sit in
packed tiles -- the corridor sets of BASELINE config 5 -- differentiate rayen/constraint_module.py:351-474 with every
product on v_mfma_f32_32x32x16_f16.  The analogue of test_gpu_parity.py::test_split_operand_kernels_are_fp32_grade for
gradients: against the fp64 backward on the SAME kappa / arg-max record its error must be that of fp32 arithmetic (no
worse than the exact-fp32 matrix-core backward it replaces, x2 with a floor at the rounding level), the creation-time
measurement that admitted the pack is asserted, and every addressing mode of the rows is covered.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from rayen_amd import ops, pack as _pack, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _sets():
    rng = np.random.default_rng(5)
    # (NA_E = I sets with FEW quadratics keep the dense-form backward, rayen_mfma_bwd.hip; with more than 64 of them
    # that kernel declines and the packed tiles serve)
    packed_identity = workloads.random_lin_quad_soc(k=32, m=64, n_quad=0, n_soc=0, seed=63)  # NA_E = I, 70 packed factors
    for i in range(70):
        U = rng.uniform(-1, 1, size=(1 + i % 7, 32))                                       # (+1 from the linear term: 2..8)
        packed_identity["P"].append(U.T @ U)
        packed_identity["q"].append(rng.uniform(-1, 1, size=(32, 1)))
        packed_identity["r"].append(rng.uniform(-1, 0, size=(1, 1)))
    ragged_identity = workloads.random_lin_quad_soc(k=22, m=40, n_quad=0, n_soc=0, seed=66)  # NA_E = I, n = 22
    for i in range(66):
        U = rng.uniform(-1, 1, size=(1 + i % 5, 22)) * 10.0 ** rng.integers(-3, 3)           # sizes over six decades
        ragged_identity["P"].append(U.T @ U)
        ragged_identity["q"].append(rng.uniform(-1, 1, size=(22, 1)) * np.abs(U).max())
        ragged_identity["r"].append(rng.uniform(-1, 0, size=(1, 1)))
    return {
        "c5r": workloads.make_raw("c5r", seed=21),                                            # n = 30 of k = 45
        "c5": workloads.make_raw("c5", seed=0),                                               # the corridor set
        "eq_n20": workloads.corridor_like(k=28, n_eq=8, m=120, n_quad=10, rank=3, seed=31),   # n = 20 of k = 28 (one block of g)
        "eq_k64": workloads.corridor_like(k=64, n_eq=34, m=90, n_quad=12, rank=5, seed=32),   # n = 30 of k = 64, rank 5: both halves
        "packed_identity": packed_identity,
        "ragged_identity": ragged_identity,
    }


def _layers(raw):
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        layer64 = ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    return cs, layer, layer64


def _row_err(got, want):
    size = np.maximum(np.abs(want).max(axis=1), 1e-30)
    return np.abs(got - want).max(axis=1) / size


@pytest.mark.parametrize("name", ["c5r", "c5", "eq_n20", "eq_k64", "packed_identity", "ragged_identity"])
def test_pair_backward_is_fp32_grade(name):
    cs, layer, layer64 = _layers(_sets()[name])
    dev = torch.device("cuda", 0)
    dp, _ = layer.device_pack(dev)
    info = dp.info()
    assert info.bwd_f32 == 3, (name, info.bwd_f32, info.bwd32_check_pair, info.bwd32_check_exact)
    # the creation-time measurement that admitted the pack
    assert 0.0 <= info.bwd32_check_pair <= max(4e-6, 1.5 * info.bwd32_check_exact)
    exact = _pack.DevicePack(layer.packed_constants(), 0, fp32_mode=1)            # the exact-fp32 kernels, forward and backward
    assert exact.info().bwd_f32 == 2
    dp64, _ = layer64.device_pack(dev)
    B = 20000 + 37
    gen = torch.Generator().manual_seed(17)
    scale = 1.5 if name != "c5" else 3.0
    v = torch.empty(B, cs.n).uniform_(-scale, scale, generator=gen)
    v[:64] *= 1e-4                                                               # interior: the gradient is NA_E' g
    v[64:96] *= 40.0                                                             # far outside
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
    g[100:164] *= 2.0 ** (-torch.randint(0, 16, (64, cs.k), generator=gen).float())   # components over 16 binades
    g[170] = 0.0
    v, g = v.cuda(), g.cuda()
    # one record for everybody: the fp64 forward's
    _, kappa64, active = ops.project_raw(v.double(), dp64, want_active=True)
    kappa = kappa64.float()
    sure = ((kappa64 - 1.0).abs() > 1e-5).cpu().numpy()                          # (kappa within rounding of 1: a kink)
    truth = ops.backward_raw(v.double(), kappa64, active, g.double(), dp64).cpu().numpy()
    got = ops.backward_raw(v, kappa, active, g, dp).cpu().double().numpy()
    ref = ops.backward_raw(v, kappa, active, g, exact, bucketed=False).cpu().double().numpy()
    lane = ops.backward_raw(v, kappa, active, g, dp, force_generic=True).cpu().double().numpy()
    assert np.all(np.isfinite(got))
    e_pair, e_exact, e_lane = (_row_err(a, truth)[sure] for a in (got, ref, lane))
    assert e_pair.max() <= max(2.0 * max(e_exact.max(), e_lane.max()), 5e-7), (e_pair.max(), e_exact.max(), e_lane.max())
    assert e_pair.mean() <= max(2.0 * e_exact.mean(), 1e-7), (e_pair.mean(), e_exact.mean())
    # interior samples and the zero gradient row
    inside = (kappa64[:64] < 1.0).cpu().numpy()
    assert inside.sum() >= 8
    lift = g[:64].cpu().double().numpy() @ np.asarray(cs.NA_E)
    assert np.max(np.abs(got[:64] - lift)[inside]) <= 1e-6 * max(1.0, np.max(np.abs(lift)))
    assert np.all(got[170] == 0.0)


@pytest.mark.parametrize("name", ["c5r", "eq_n20", "ragged_identity"])
@pytest.mark.parametrize("B", [1, 31, 64, 1000, 4096 + 5])
def test_pair_backward_addressing_modes(name, B):
    """Rows back to back behind aligned bases (whole 16-byte pieces of the group's block), rows at a padded leading
    dimension, and a base that is not 16-byte aligned: the same gradient, bit for bit; nothing written outside."""
    cs, layer, _ = _layers(_sets()[name])
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    assert dp.info().bwd_f32 == 3
    gen = torch.Generator().manual_seed(B)
    v = torch.empty(B, cs.n).uniform_(-1.5, 1.5, generator=gen).cuda()
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen).cuda()
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    flat = ops.backward_raw(v, kappa, active, g, dp)
    # padded leading dimension of v (and so of grad_v): columns beyond n stay zero
    wide = torch.zeros(B, cs.n + 6, device="cuda")
    wide[:, :cs.n] = v
    vv = wide[:, :cs.n]
    assert vv.stride(0) == cs.n + 6
    got = ops.backward_raw(vv, kappa, active, g, dp)
    assert torch.equal(got[:, :cs.n].contiguous(), flat)
    # mis-aligned bases
    buf = torch.empty(B * cs.n + 4, device="cuda")
    v2 = buf[1:1 + B * cs.n].view(B, cs.n)
    v2.copy_(v)
    gbuf = torch.empty(B * cs.k + 4, device="cuda")
    g2 = gbuf[3:3 + B * cs.k].view(B, cs.k)
    g2.copy_(g)
    assert v2.data_ptr() % 16 != 0 and g2.data_ptr() % 16 != 0
    assert torch.equal(ops.backward_raw(v2, kappa, active, g2, dp), flat)
    # and against the lane-per-sample backward (same record, same branch)
    lane = ops.backward_raw(v, kappa, active, g, dp, force_generic=True)
    size = lane.abs().amax(1).clamp_min(1e-20)
    assert float(((flat - lane).abs().amax(1) / size).max()) <= 2e-4


def test_modes_pin_the_backward_family():
    """fp32_mode 1 / 4 never build the f16-pair backward, 3 takes it unmeasured, 0 measures it (the forward's switch)."""
    cs, layer, _ = _layers(_sets()["c5r"])
    consts = layer.packed_constants()
    assert _pack.DevicePack(consts, 0, fp32_mode=1).info().bwd_f32 == 2
    assert _pack.DevicePack(consts, 0, fp32_mode=4).info().bwd_f32 == 2
    forced = _pack.DevicePack(consts, 0, fp32_mode=3).info()
    assert forced.bwd_f32 == 3 and forced.bwd32_check_pair == -1.0
    measured = _pack.DevicePack(consts, 0).info()
    assert measured.bwd_f32 == 3 and 0.0 <= measured.bwd32_check_pair <= max(4e-6, 1.5 * measured.bwd32_check_exact)
    # other shapes keep their kernels: config 4 (LMI) 4, config 2 (n = 16) 1; config 3 (dense forms at n = k = 64) has its own
    # f16-pair backward since round 6 (7, tests/test_gpu_backward_dense_pairs.py), the exact kernel 1 under fp32_mode 1
    cs3 = workloads.build_constraints(workloads.make_raw("c3", seed=1))
    assert _pack.DevicePack(ConstraintModule(cs3, create_map=False).cuda().packed_constants(), 0, fp32_mode=1).info().bwd_f32 == 1
    for name, want in (("c3", (7,)), ("c4", (4,)), ("c2", (1,))):
        cs2 = workloads.build_constraints(workloads.make_raw(name, seed=1))
        info = ConstraintModule(cs2, create_map=False).cuda().device_pack(torch.device("cuda", 0))[0].info()
        assert info.bwd_f32 in want, (name, info.bwd_f32)
