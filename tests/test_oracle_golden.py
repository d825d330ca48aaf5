"""Pin the CPU oracle against outputs of the real reference (tests/golden/*.npz).

The fixtures were produced by ``tests/golden/make_golden.py`` importing
``/root/reference`` in the build container.  fp64 must agree to ~1e-12; fp32 to a
few ulp-level multiples (the oracle issues the same op sequence on the same
PyTorch-CPU build, so it is usually bit-identical).
"""
import numpy as np
import pytest
import torch

from helpers import golden_names, load_golden, rel_err_rows
from oracle import rayen_oracle as oracle


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("tag,dtype,tol", [("64", torch.float64, 1e-12), ("32", torch.float32, 2e-6)])
def test_oracle_matches_reference(name, tag, dtype, tol):
    raw, csd, z = load_golden(name)
    buf = oracle.precompute(csd, dtype=dtype)
    for bname in ("D", "all_phi", "all_delta", "L"):
        key = f"buf_{bname}{tag}"
        if key in z:
            ref = z[key]
            got = buf[bname].numpy()
            scale = max(1.0, float(np.max(np.abs(ref))))
            assert np.max(np.abs(got - ref)) <= tol * scale, bname
    x = torch.tensor(z["x"]).to(dtype)
    y = oracle.forward(buf, x, check_nan=False).numpy()[:, :, 0]
    assert y.shape == z["y" + tag].shape
    # rows on which the reference's own fp32 arithmetic is NaN (helpers.load_golden): the oracle, issuing the same op
    # sequence, must be NaN on exactly those rows -- and nowhere else, at either precision
    ok = ~z["nan_rows32"] if tag == "32" else np.ones(len(y), dtype=bool)
    assert np.array_equal(np.isfinite(y).all(axis=1), ok)
    assert np.max(rel_err_rows(y[ok], z["y" + tag][ok])) <= tol
    n = csd["NA_E"].shape[1]
    v_bar = torch.nn.functional.normalize(x[:, 0:n, 0:1], dim=1)
    kappa = oracle.compute_kappa(buf, v_bar).numpy()[:, 0, 0]
    ref_k = z["kappa_bar" + tag]
    assert np.array_equal(np.isfinite(kappa), ok)
    assert np.max(np.abs(kappa[ok] - ref_k[ok]) / np.maximum(np.abs(ref_k[ok]), 1.0)) <= tol


@pytest.mark.parametrize("name", golden_names())
def test_reference_outputs_are_feasible(name):
    """Sanity of the fixtures themselves: the reference's fp64 outputs satisfy the constraints."""
    raw, csd, z = load_golden(name)
    assert oracle.max_violation(raw, z["y64"]) < 1e-9
    assert oracle.max_violation(raw, z["y32"]) < 1e-4


def test_interior_rows_are_unclipped():
    """v=0 maps to y0 and a tiny step stays unclipped (rows 500/501 of the example fixtures)."""
    raw, csd, z = load_golden("example_02")
    y0 = raw["y0"][:, 0]
    assert np.allclose(z["y64"][500], y0)
    step = z["x"][501, :, 0]
    assert np.allclose(z["y64"][501], y0 + csd["NA_E"] @ step, atol=1e-12)


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("tag,dtype,tol", [("64", torch.float64, 1e-12), ("32", torch.float32, 2e-6)])
def test_oracle_rayen_old_matches_reference(name, tag, dtype, tol):
    raw, csd, z = load_golden(name)
    buf = oracle.precompute(csd, dtype=dtype)
    x = torch.cat((torch.tensor(z["x"]), torch.tensor(z["beta"])), dim=1).to(dtype)
    y = oracle.forward(buf, x, method="RAYEN_old", check_nan=False).numpy()[:, :, 0]
    ok = ~z["nan_rows32"] if tag == "32" else np.ones(len(y), dtype=bool)
    assert np.array_equal(np.isfinite(y).all(axis=1), ok)
    assert np.max(rel_err_rows(y[ok], z["y_old" + tag][ok])) <= tol
