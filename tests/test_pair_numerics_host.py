"""Host-side emulation of the f16-pair operand format (rayen_amd/csrc/rayen_mfma_pair.hip): the pieces are formed as
the kernel and its image builder form them (power-of-two scale, round to nearest f16, exact remainder), the three
piece products are summed in fp64.  What the kernel's accuracy rests on: the row results T = W v are as close to the
exact ones as an fp32 FMA chain's -- representation errors do not accumulate along K.  No GPU."""
import numpy as np
import pytest
import torch

from rayen_amd import workloads
from rayen_amd.constraint_module import ConstraintModule


def _pair(x, scale):
    xs = x * scale
    x1 = xs.astype(np.float16).astype(np.float64)
    x2 = (xs - x1).astype(np.float16).astype(np.float64)
    return x1 / scale, x2 / scale


def _bf16(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).astype(np.float64)


@pytest.mark.parametrize("name", ["c2", "c3", "c5"])
def test_row_results_of_the_pair_format_are_fp32_grade(name):
    cs = workloads.build_constraints(workloads.make_raw(name, seed=21))
    consts = ConstraintModule(cs, create_map=False).packed_constants()
    rng = np.random.default_rng(0)
    span = workloads.CONFIGS[name][3]
    v = rng.uniform(-span, span, size=(4000, cs.n)).astype(np.float32).astype(np.float64)
    v[:400] *= 10.0 ** rng.integers(-6, 7, size=(400, 1))                      # rows of very different magnitudes
    W = consts.W.astype(np.float32).astype(np.float64)                         # what the device holds
    exact = v @ W.T
    size = np.abs(v) @ np.abs(W).T + 1e-300
    g_w = 2.0 ** (13 - np.floor(np.log2(np.abs(W).max())))                     # largest entry into [2^13, 2^14)
    W1, W2 = _pair(W, g_w)
    s_v = 2.0 ** (13 - np.floor(np.log2(np.abs(v).max(axis=1, keepdims=True))))
    v1, v2 = _pair(v, s_v)
    assert np.abs(W - W1 - W2).max() <= 2.0 ** -22 * np.abs(W).max()
    assert (np.abs(v - v1 - v2).max(axis=1) <= 2.0 ** -22 * np.abs(v).max(axis=1)).all()
    pair = v1 @ W1.T + (v1 @ W2.T + v2 @ W1.T)
    chain = (v.astype(np.float32) @ W.T.astype(np.float32)).astype(np.float64)  # fp32 accumulation
    b1 = _bf16(W); b2 = _bf16(W - b1); b3 = _bf16(W - b1 - b2)
    c1 = _bf16(v); c2 = _bf16(v - c1); c3 = _bf16(v - c1 - c2)
    triple = c1 @ b1.T + (c1 @ b2.T + c2 @ b1.T) + (c1 @ b3.T + c2 @ b2.T + c3 @ b1.T)
    e_pair, e_chain, e_triple = (np.abs(t - exact) / size for t in (pair, chain, triple))
    assert e_pair.max() <= 4e-7                                                # ~2^-22, not K times that
    assert e_pair.max() <= 2.0 * e_chain.max()                                 # no worse than fp32 arithmetic
    assert e_triple.max() <= 5e-8                                              # the six-product scheme: ~2^-24 terms only


def test_second_piece_stays_in_f16_range():
    """With the largest component at 2^13..2^14 the remainder of every component down to 2^-17 of the largest is a
    NORMAL f16 (full 11 bits); below that the error is bounded by the f16 subnormal spacing, 2^-24 of the scaled value
    = 2^-38 of the largest component."""
    x = 2.0 ** np.arange(13.9, -24.0, -0.37)
    x1 = x.astype(np.float16).astype(np.float64)
    x2 = (x - x1).astype(np.float16).astype(np.float64)
    err = np.abs(x - x1 - x2)
    big = x >= 2.0 ** (13.9 - 17)
    assert (err[big] <= 2.0 ** -22 * x[big]).all()
    assert (err <= 2.0 ** -25 + 2.0 ** -22 * x).all()
    assert np.isfinite(x1).all() and x1.max() < 65504
