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


def test_backward_of_a_wide_set_detours_loudly_and_matches_autograd():
    """n = 400 with quadratics and cones: the forward runs on the lane kernels; a backward they do not stage comes
    from autograd through the packed evaluator, with a warning, and equals autograd through the reference's ops."""
    raw = workloads.random_lin_quad_soc(k=400, m=60, n_quad=2, n_soc=1, r_M=30, seed=9)
    cs, layer = _layer64(raw)
    x = torch.empty(48, cs.n, 1, dtype=torch.float64).uniform_(-1, 1)
    w = torch.empty(48, cs.k, 1, dtype=torch.float64).uniform_(-1, 1)
    xg = x.cuda().requires_grad_(True)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        (layer(xg) * w.cuda()).sum().backward()
    x2 = x.clone().requires_grad_(True)
    (oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x2) * w).sum().backward()
    err = (xg.grad.cpu() - x2.grad).abs().amax(dim=(1, 2)) / x2.grad.abs().amax(dim=(1, 2)).clamp_min(1e-30)
    # (kinks -- ties of the arg-max, kappa = 1 -- are measure-zero for these random directions at fp64)
    assert float(err.max()) <= 1e-7
