#!/usr/bin/env python
"""Developer timing: the W-in-LDS schedule against the other schedules on n = 32 sets at small batches
    RAYEN_WL_MIN_GROUPS=1|100000000 python scripts/ubench/wl_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rayen_amd import _lib, ops, workloads
from rayen_amd.constraint_module import ConstraintModule
sets = {"n32 (300 rows, 3 quad, 2 soc)": workloads.random_lin_quad_soc(k=32, m=300, n_quad=3, n_soc=2, seed=17),
        "n32 tiny (32 rows, 1 quad)": workloads.random_lin_quad_soc(k=32, m=32, n_quad=1, n_soc=0, seed=3),
        "n64 small (64 rows, 1 quad)": workloads.random_lin_quad_soc(k=64, m=64, n_quad=1, n_soc=0, seed=4)}
for name, raw in sets.items():
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    out = []
    for B in (32, 1024, 4096, 16384, 65536):
        x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
        y = torch.empty(B, cs.k, device="cuda")
        call = lambda: ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
        g = torch.cuda.CUDAGraph()
        for _ in range(20): call()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20): call()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): g.replay()
        e1.record(); torch.cuda.synchronize()
        out.append(f"B={B}: {e0.elapsed_time(e1) / 400 * 1e3:.1f}")
    print(name, "| kernel", _lib.load().rayen_last_forward_kernel(), "| us per launch (graph of 20):", "  ".join(out), flush=True)
