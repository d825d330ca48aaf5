"""The product's own torch evaluator of the packed form (``rayen_amd/eager.py``) -- what serves tensors on the host.

Held to the reference's golden vectors (``tests/golden/*.npz``) at the GPU path's own bars (1e-5 fp32, 1e-12-grade
fp64), to the oracle's autograd for its gradients, and to the reference's CPU smoke usage
(``/root/reference/examples/test_layer.py:70-117``: build the layer, feed ``U(-5, 5)`` samples, check feasibility).
It must not import ``oracle/`` (the test checks the source), and a HIP tensor must never reach it silently.
"""
import os
import warnings

import numpy as np
import pytest
import torch

from helpers import csd_from_cs, golden_names, load_golden, rel_err_rows
from oracle import rayen_oracle as oracle
from rayen_amd import constraints, eager, workloads
from rayen_amd.constraint_module import ConstraintModule

FP32_TOL = 1e-5
FP64_TOL = 1e-11


def _layer(raw_or_cs, dtype, method="RAYEN", **kw):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = raw_or_cs if isinstance(raw_or_cs, constraints.ConvexConstraints) else workloads.build_constraints(raw_or_cs)
        kw.setdefault("create_map", False)
        return cs, ConstraintModule(cs, method=method, **kw)
    finally:
        torch.set_default_dtype(prev)


def _to_my_basis(cs, csd_ref, x, dtype):
    R = cs.NA_E.T @ csd_ref["NA_E"]
    return torch.tensor(x[:, :, 0].astype(np.float64) @ R.T, dtype=dtype).unsqueeze(2)


def test_the_evaluator_does_not_touch_the_oracle():
    src = open(eager.__file__).read()
    code = [ln for ln in src.splitlines() if ln.strip().startswith(("import ", "from "))]
    assert not any("oracle" in ln for ln in code), code
    assert "/root/reference" not in src and "subprocess" not in src


@pytest.mark.parametrize("name", golden_names())
@pytest.mark.parametrize("tag,dtype,tol", [("32", torch.float32, FP32_TOL), ("64", torch.float64, FP64_TOL)])
def test_golden_on_cpu_tensors(name, tag, dtype, tol):
    raw, csd, z = load_golden(name)
    cs, layer = _layer(raw, dtype)
    x = _to_my_basis(cs, csd, z["x"], dtype)
    y = layer(x)
    assert y.device.type == "cpu" and y.dtype == dtype and tuple(y.shape) == (x.shape[0], cs.k, 1)
    y = y.numpy()[:, :, 0]
    assert np.max(rel_err_rows(y, z["y" + tag])) <= tol
    floor = 1e-6 if tag == "32" else 1e-11
    assert oracle.max_violation(raw, y) <= max(floor, 3 * oracle.max_violation(raw, z["y" + tag]))
    v_bar = torch.nn.functional.normalize(x[:, 0:cs.n, 0:1], dim=1)
    kb = layer.computeKappa(v_bar).numpy()[:, 0, 0]
    ref = z["kappa_bar" + tag]
    assert np.max(np.abs(kb - ref) / np.maximum(1.0, np.abs(ref))) <= (2 * tol if tag == "32" else 1e-8)


@pytest.mark.parametrize("name", ["example_00", "example_13", "config_c2", "config_c4"])
@pytest.mark.parametrize("tag,dtype,tol", [("32", torch.float32, FP32_TOL), ("64", torch.float64, FP64_TOL)])
def test_golden_old_head_on_cpu_tensors(name, tag, dtype, tol):
    raw, csd, z = load_golden(name)
    cs, layer = _layer(raw, dtype, method="RAYEN_old")
    x = torch.cat((_to_my_basis(cs, csd, z["x"], dtype), torch.tensor(z["beta"]).to(dtype)), dim=1)
    y = layer(x).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y, z["y_old" + tag])) <= tol


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5r"])
def test_gradients_match_autograd_through_the_oracle(name):
    """d(sum w.y)/dx through the evaluator against autograd through the reference's op sequence, fp64, away from kinks."""
    from helpers import kink_mask
    cs, layer = _layer(workloads.make_raw(name, seed=5), torch.float64)
    gen = torch.Generator().manual_seed(2)
    x = torch.empty(96, cs.n, 1, dtype=torch.float64).uniform_(-2, 2, generator=gen).requires_grad_(True)
    w = torch.empty(96, cs.k, 1, dtype=torch.float64).uniform_(-1, 1, generator=gen)
    (layer(x) * w).sum().backward()
    got = x.grad.clone()
    x2 = x.detach().clone().requires_grad_(True)
    buf = oracle.precompute(csd_from_cs(cs), torch.float64)
    (oracle.forward(buf, x2) * w).sum().backward()
    keep = ~kink_mask(oracle, cs, x.detach(), 1e-6)
    assert keep.sum() > 48
    err = (got - x2.grad)[keep].abs().amax(dim=(1, 2)) / x2.grad[keep].abs().amax(dim=(1, 2)).clamp_min(1e-30)
    assert float(err.max()) <= 1e-7


def test_reference_smoke_usage_on_the_host():
    """examples/test_layer.py:70-117 in spirit: every example set, CPU tensors, mapper in front, outputs feasible."""
    torch.manual_seed(0)
    for index in (0, 2, 4, 9, 13):
        raw, _, _ = load_golden(f"example_{index:02d}")
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, input_dim=7, method="RAYEN", create_map=True)
        x = torch.empty(500, 7, 1).uniform_(-5.0, 5.0)
        y = layer(x)
        assert tuple(y.shape) == (500, cs.k, 1)
        assert cs.getMaxViolation(y.detach().double().numpy()[:, :, 0]) < 1e-5
        y.sum().backward()
        assert layer.mapper.weight.grad is not None and torch.isfinite(layer.mapper.weight.grad).all()


def test_sixteen_bit_inputs_are_computed_in_fp32():
    cs, layer = _layer(workloads.make_raw("c2", seed=1), torch.float32)
    x = torch.empty(64, cs.n, 1).uniform_(-1, 1)
    y16 = layer(x.to(torch.bfloat16))
    assert y16.dtype == torch.bfloat16
    y = layer(x.to(torch.bfloat16).float())
    assert torch.equal(y.to(torch.bfloat16), y16)


def test_empty_batch_and_nan_assert():
    cs, layer = _layer(workloads.make_raw("c2", seed=1), torch.float32)
    assert tuple(layer(torch.empty(0, cs.n, 1)).shape) == (0, cs.k, 1)
    x = torch.zeros(3, cs.n, 1)
    x[1, 0, 0] = float("nan")
    with pytest.raises(AssertionError):
        layer(x)


def test_state_follows_to_and_double():
    cs, layer = _layer(workloads.make_raw("c2", seed=1), torch.float32)
    x = torch.empty(32, cs.n, 1).uniform_(-1, 1)
    y32 = layer(x)
    layer = layer.double()
    y64 = layer(x.double())
    assert y64.dtype == torch.float64
    assert float((y64 - y32.double()).abs().max()) < 1e-5
