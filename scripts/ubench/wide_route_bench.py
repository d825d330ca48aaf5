#!/usr/bin/env python
"""Developer timing of the wide route (vendor GEMM + products epilogue, rayen_wide.hip) against the lane-per-sample
kernel on shapes of the reference's sweep (examples/scripts/time_analysis.py:57-130, B = 2000), and against the
reference's op sequence on the host for the smaller ones:   python scripts/ubench/wide_route_bench.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from rayen_amd import _lib, ops, workloads                  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule    # noqa: E402


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 2000
for label, kw in (("lin r=1000 k=1000", dict(k=1000, m=1000, n_quad=0, n_soc=0)),
                  ("lin r=3000 k=5000", dict(k=5000, m=3000, n_quad=0, n_soc=0)),
                  ("qp eta=10 k=500", dict(k=500, m=0, n_quad=10, n_soc=0)),
                  ("qp eta=50 k=300", dict(k=300, m=0, n_quad=50, n_soc=0)),
                  ("soc mu=100 r_M=100 k=500", dict(k=500, m=0, n_quad=0, n_soc=100, r_M=100)),
                  ("mixed k=256: 512 lin + 8 qp + 8 soc", dict(k=256, m=512, n_quad=8, n_soc=8))):
    raw = workloads.random_lin_quad_soc(seed=1, **kw)
    t0 = time.time()
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    setup = time.time() - t0
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    y, kap, act = ops.project_raw(v, dp)
    assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PRODUCTS
    wide = t(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False))
    lane = t(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False, force_generic=True), reps=3)
    yl, _, _ = ops.project_raw(v, dp, force_generic=True)
    out = {"set": label, "B": B, "n": cs.n, "rows_of_W": int(dp.consts.W.shape[0]), "setup_s": round(setup, 2),
           "wide_route_ms": round(wide, 4), "lane_kernel_ms": round(lane, 4), "speedup": round(lane / wide, 1),
           "max_abs_diff_vs_lane": float((y - yl).abs().max()), "clipped": float((kap > 1).float().mean()),
           "max_violation": float(cs.getMaxViolation(y[:256].cpu().double().numpy()))}
    print(json.dumps(out), flush=True)
