#!/usr/bin/env python
"""Developer timing: the fp32 backward of small sets at small batches, bucketed (count + scatter + walk) against the plain walk
(one launch):  python scripts/ubench/bwd_small.py [config ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule


def t(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name in sys.argv[1:] or ["c2", "c5r"]:
    cs = workloads.build_constraints(workloads.make_raw(name, seed=0))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    out = []
    for B in (1024, 4096, 16384, 65536, 262144):
        v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5)
        g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
        _, kappa, active = ops.project_raw(v, dp, want_active=True)
        a = t(lambda: ops.backward_raw(v, kappa, active, g, dp))
        b = t(lambda: ops.backward_raw(v, kappa, active, g, dp, bucketed=False))
        out.append(f"B={B}: bucketed {a:.1f} plain {b:.1f}")
    print(name, "bwd_f32 =", dp.info().bwd_f32, "| us per call:", "  ".join(out), flush=True)
