"""The fp32 backward of dense-form sets at n = k = 64 on f16 pairs (rayen_amd/csrc/rayen_mfma_bwdd.hip, round 6): config 3's
training step differentiates rayen/constraint_module.py:351-474 in ONE launch -- S_s v of every quadratic / cone for every
sample on v_mfma_f32_32x32x16_f16, the forms resident in LDS, the batch streamed in order -- instead of the bucketed
exact-fp32 walk (count + scatter + walk).  Against the fp64 backward on the SAME kappa / arg-max record its error must be
that of fp32 arithmetic (no worse than 2 x the exact-fp32 matrix-core backward it replaces, with a floor at the rounding
level); the creation-time measurement that admitted the pack is asserted; ragged batches, padded leading dimensions,
interior samples, every kind of active constraint, and the shapes it must decline.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from rayen_amd import ops, pack as _pack, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _sets():
    return {
        "c3": workloads.make_raw("c3", seed=7),
        "c3_other_seed": workloads.make_raw("c3", seed=3),
        "quads_only": workloads.random_lin_quad_soc(k=64, m=96, n_quad=6, n_soc=0, seed=41),
        "cones_only": workloads.random_lin_quad_soc(k=64, m=64, n_quad=0, n_soc=3, seed=42),
        # n = k below 64 in whole 16-byte pieces (round 6): tiles padded to 64 columns, the pieces beyond n addressed out of range
        "n60": workloads.random_lin_quad_soc(k=60, m=96, n_quad=3, n_soc=2, seed=43),
        "n36": workloads.random_lin_quad_soc(k=36, m=64, n_quad=2, n_soc=1, seed=44),
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


@pytest.mark.parametrize("name", ["c3", "c3_other_seed", "quads_only", "cones_only", "n60", "n36"])
def test_dense_pair_backward_is_fp32_grade(name):
    cs, layer, layer64 = _layers(_sets()[name])
    dev = torch.device("cuda", 0)
    dp, _ = layer.device_pack(dev)
    info = dp.info()
    assert info.bwd_f32 == 7, (name, info.bwd_f32, info.bwd32_check_pair, info.bwd32_check_exact)
    assert 0.0 <= info.bwd32_check_pair <= max(4e-6, 1.5 * info.bwd32_check_exact)
    exact = _pack.DevicePack(layer.packed_constants(), 0, fp32_mode=1)            # the exact-fp32 kernels, forward and backward
    assert exact.info().bwd_f32 == 1
    dp64, _ = layer64.device_pack(dev)
    B = 65536 + 4096 + 37
    gen = torch.Generator().manual_seed(17)
    v = torch.empty(B, cs.n).uniform_(-1.5, 1.5, generator=gen)
    v[:64] *= 1e-4                                                               # interior: the gradient is g itself
    v[64:96] *= 40.0                                                             # far outside
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
    g[100:164] *= 2.0 ** (-torch.randint(0, 16, (64, cs.k), generator=gen).float())   # components over 16 binades
    g[170] = 0.0
    v, g = v.cuda(), g.cuda()
    _, kappa64, active = ops.project_raw(v.double(), dp64, want_active=True)      # one record for everybody: the fp64 forward's
    kappa = kappa64.float()
    sure = ((kappa64 - 1.0).abs() > 1e-5).cpu().numpy()                          # (kappa within rounding of 1: a kink)
    truth = ops.backward_raw(v.double(), kappa64, active, g.double(), dp64).cpu().numpy()
    got = ops.backward_raw(v, kappa, active, g, dp).cpu().double().numpy()
    ref = ops.backward_raw(v, kappa, active, g, exact, bucketed=False).cpu().double().numpy()
    assert np.all(np.isfinite(got))
    e_pair, e_exact = (_row_err(a, truth)[sure] for a in (got, ref))
    print(f"\n  [{name}] worst gradient row against fp64: f16 pairs {e_pair.max():.3e} (mean {e_pair.mean():.2e}), exact fp32 "
          f"{e_exact.max():.3e} (mean {e_exact.mean():.2e}); creation-time probe {info.bwd32_check_pair:.2e} / {info.bwd32_check_exact:.2e}")
    assert e_pair.max() <= max(2.0 * e_exact.max(), 5e-7), (e_pair.max(), e_exact.max())
    assert e_pair.mean() <= max(2.0 * e_exact.mean(), 1e-7), (e_pair.mean(), e_exact.mean())
    # every kind of active constraint took part
    segs = active[:, 0].cpu().numpy()[(kappa64 > 1).cpu().numpy()]
    assert len(np.unique(segs)) >= 2
    # interior samples and the zero gradient row
    inside = (kappa64[:64] < 1.0).cpu().numpy()
    assert inside.sum() >= 8
    assert np.array_equal(got[:64][inside], g[:64].cpu().double().numpy()[inside])
    assert np.all(got[170] == 0.0)


@pytest.mark.parametrize("name", ["c3", "n36"])
@pytest.mark.parametrize("B", [65536, 65536 + 31, 100001, 262144])
def test_dense_pair_backward_equals_itself_in_every_addressing_mode(B, name):
    """Padded leading dimensions of v, grad_y and grad_v: the same gradient bit for bit, nothing written beyond the n columns or
    beyond the batch (the last group is ragged; at n = 36 the pieces beyond a row's columns are out of range)."""
    cs, layer, _ = _layers(_sets()[name])
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    assert dp.info().bwd_f32 == 7
    gen = torch.Generator().manual_seed(B)
    v = torch.empty(B, cs.n).uniform_(-1.5, 1.5, generator=gen).cuda()
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen).cuda()
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    flat = ops.backward_raw(v, kappa, active, g, dp)
    wide_v = torch.zeros(B, cs.n + 8, device="cuda")
    wide_v[:, :cs.n] = v
    wide_g = torch.zeros(B, cs.k + 4, device="cuda")
    wide_g[:, :cs.k] = g
    wide_v[:, cs.n:] = 123.0                       # (columns beyond n: never read as data ...)
    wide_g[:, cs.k:] = -55.0
    got = ops.backward_raw(wide_v[:, :cs.n], kappa, active, wide_g[:, :cs.k], dp)
    assert torch.equal(got[:, :cs.n], flat)
    # against the lane-per-sample backward (another instruction stream, exact fp32): fp32-grade agreement
    lane = ops.backward_raw(v, kappa, active, g, dp, force_generic=True)
    size = lane.abs().amax(dim=1).clamp_min(1e-30)
    assert float(((flat - lane).abs().amax(dim=1) / size).max()) <= 2e-5


def test_small_batches_are_served_too_and_misaligned_rows_stay_on_the_exact_kernel():
    """Round 6: no batch threshold (one workgroup per CU as soon as there is a group for it: 0.012 ms at B = 1 024 against 0.072
    for the bucketed walk) -- a slice of a batch gives the slice of its gradient bit for bit (rows are independent); rows that
    are not 16-byte aligned cannot be read as pieces: the bucketed exact-fp32 kernel serves them, same gradients to fp32 rounding."""
    cs, layer, _ = _layers(_sets()["c3"])
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(5)
    B = 70000
    v = torch.empty(B, cs.n).uniform_(-1.5, 1.5, generator=gen).cuda()
    g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen).cuda()
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    big = ops.backward_raw(v, kappa, active, g, dp)
    for lo, hi in ((0, 4096), (0, 1), (33, 64), (5000, 5000 + 1037)):
        small = ops.backward_raw(v[lo:hi].clone(), kappa[lo:hi].clone(), active[lo:hi].clone(), g[lo:hi].clone(), dp)
        assert torch.equal(small, big[lo:hi]), (lo, hi)
    exact = _pack.DevicePack(layer.packed_constants(), 0, fp32_mode=1)
    ref = ops.backward_raw(v[:4096], kappa[:4096], active[:4096], g[:4096], exact)
    size = ref.abs().amax(dim=1).clamp_min(1e-30)
    assert float(((big[:4096] - ref).abs().amax(dim=1) / size).max()) <= 2e-5
    buf = torch.empty(B * cs.n + 4, device="cuda")
    w = buf[1:1 + B * cs.n].view(B, cs.n)
    w.copy_(v)
    assert w.data_ptr() % 16 != 0
    off = ops.backward_raw(w, kappa, active, g, dp)
    size = big.abs().amax(dim=1).clamp_min(1e-30)
    assert float(((off - big).abs().amax(dim=1) / size).max()) <= 2e-5


def test_training_step_of_the_module_runs_on_it():
    """autograd through ConstraintModule.forward at the headline shape: the gradient of a scalar loss with respect to the
    input agrees with the fp64 module's (same inputs) to fp32 rounding on the rows whose arg-max record agrees."""
    cs, layer, layer64 = _layers(_sets()["c3"])
    B = 131072
    x = torch.empty(B, cs.n, 1, device="cuda").uniform_(-1.5, 1.5, generator=torch.Generator(device="cuda").manual_seed(9))
    w = torch.empty(B, cs.k, 1, device="cuda").uniform_(-1, 1, generator=torch.Generator(device="cuda").manual_seed(10))
    x32 = x.clone().requires_grad_(True)
    (layer(x32) * w).sum().backward()
    x64 = x.double().requires_grad_(True)
    (layer64(x64) * w.double()).sum().backward()
    g32, g64 = x32.grad[:, :, 0].double(), x64.grad[:, :, 0]
    size = g64.abs().amax(dim=1).clamp_min(1e-30)
    err = (g32 - g64).abs().amax(dim=1) / size
    # (rows on a kink -- kappa within rounding of 1, or two constraints within rounding of each other -- may differentiate
    # another branch in fp32: a handful of 131 072)
    assert float(err.median()) <= 1e-6
    assert int((err > 1e-4).sum()) <= B // 1000
