"""The inward bias (round 6; RAYEN_PREPARE_INWARD_BIAS, include/rayen_hip.h; SURVEY.md section 7 "fp32 feasibility", CM:374):
with it the fp32 images of a pack evaluate (1 + 2^-20) kappa, so a clipped sample stops 9.5e-7 of its step short of the boundary
instead of ON it, where half of the fp32 roundings fall outside.  Off by default (the outputs are then the reference's);
``ConstraintModule.inward_bias = True`` before the first forward turns it on.  Needs an MI355X."""
import numpy as np
import pytest
import torch

from rayen_amd import workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _outputs(cs, x, bias, dtype=torch.float32):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ConstraintModule(cs, method="RAYEN", create_map=False)
    finally:
        torch.set_default_dtype(prev)
    layer.inward_bias = bias
    layer = layer.cuda()
    y = layer(x.to(dtype).cuda())[:, :, 0]
    kappa = layer.computeKappa(x.to(dtype).cuda())[:, 0, 0]
    assert not layer._hip_unsupported
    return y.double().cpu().numpy(), kappa.double().cpu().numpy()


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5r"])
def test_the_bias_moves_clipped_samples_inward_and_nothing_else(name):
    raw = workloads.make_raw(name, seed=0)
    cs = workloads.build_constraints(raw)
    rng = workloads.CONFIGS[name][3]
    B = 8192
    x = torch.empty(B, cs.n, 1).uniform_(-rng, rng, generator=torch.Generator().manual_seed(4))
    x[:64] *= 1e-3                                                       # interior: never clipped
    y_plain, k_plain = _outputs(cs, x, False)
    y_bias, k_bias = _outputs(cs, x, True)
    center = np.asarray(cs.y0, dtype=np.float64)[:, 0]
    step_plain = np.linalg.norm(y_plain - center, axis=1)
    step_bias = np.linalg.norm(y_bias - center, axis=1)
    clipped = k_plain > 1.0 + 1e-5
    interior = k_plain < 1.0 - 1e-5
    assert clipped.sum() > B // 20 and interior.sum() >= 64
    # interior samples: the same bits (y = y0 + NA_E v either way)
    assert np.array_equal(y_bias[interior], y_plain[interior])
    # clipped samples: the step shrinks by 2^-20 of itself (to the rounding of the fp32 arithmetic), EVERY family's -- a
    # symmetric form's rows enter under a square root and take the factor twice (c2, c3), factors / cones / generators once
    ratio = step_bias[clipped] / step_plain[clipped]
    eps = 2.0 ** -20
    stuck = ratio > 1.0 - 0.4 * eps
    print(f"\n  [{name}] clipped {clipped.sum()}, step ratio min {ratio.min():.9f} max {ratio.max():.9f}, rows that did not move: "
          f"{stuck.sum()} (kappa of those: {np.sort(k_plain[clipped][stuck])[:5]} ... {np.sort(k_plain[clipped][stuck])[-5:]})")
    # (a row does not move where 2^-20 of its step is below half an ulp of y itself: a short step from a large y0 -- c5r --,
    # or a product that happens to round the same way)
    y_size = np.abs(y_plain[clipped]).max(axis=1)
    explained = 2.0 * eps * step_plain[clipped] <= 2.0 ** -23 * y_size * np.sqrt(cs.k)
    assert ratio.max() <= 1.0 + 1e-9, ratio.max()
    assert (stuck & ~explained).sum() <= 0.002 * clipped.sum(), ((stuck & ~explained).sum(), stuck.sum())
    assert ratio.min() >= 1.0 - 6.0 * eps, ratio.min()
    assert (k_bias[clipped] < k_plain[clipped]).sum() <= 0.01 * clipped.sum()      # (kappa itself: (1 + 2^-20) x, to its own rounding)
    # feasibility: far fewer rows with a positive residual, none new above 1e-6, and parity moved by < 3e-6 of a row
    v_plain, v_bias = cs.getViolationRows(y_plain), cs.getViolationRows(y_bias)
    shift = np.abs(y_bias - y_plain).max(axis=1) / np.maximum(np.abs(y_plain).max(axis=1), 1e-30)
    print(f"\n  [{name}] rows > 0: {(v_plain > 0).sum()} -> {(v_bias > 0).sum()}, rows > 1e-6: {(v_plain > 1e-6).sum()} -> "
          f"{(v_bias > 1e-6).sum()}, max {v_plain.max():.2e} -> {v_bias.max():.2e}, shift {shift.max():.2e}")
    assert shift.max() <= 3e-6
    assert (v_bias > 1e-6).sum() <= (v_plain > 1e-6).sum()
    if cs.n == cs.k:                                   # (equality constraints leave an fp32 residual of their own on every row)
        assert (v_bias > 0).sum() <= 0.1 * (v_plain > 0).sum() + 2
        assert v_bias.max() <= max(v_plain.max(), 1e-7)


def test_fp64_is_untouched_by_the_bias():
    raw = workloads.make_raw("c3", seed=0)
    cs = workloads.build_constraints(raw)
    x = torch.empty(1024, cs.n, 1, dtype=torch.float64).uniform_(-1, 1, generator=torch.Generator().manual_seed(5))
    y_plain, _ = _outputs(cs, x, False, torch.float64)
    y_bias, _ = _outputs(cs, x, True, torch.float64)
    assert np.array_equal(y_plain, y_bias)
