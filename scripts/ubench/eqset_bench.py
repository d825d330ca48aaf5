#!/usr/bin/env python
"""Forward timings of sets with equality constraints at 32 < n <= 64 (the NKK = 2, NA_E != I instances)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule


def time_call(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for tag, kw in (("k60_eq10_q40r3", dict(k=60, n_eq=10, m=128, n_quad=40, rank=3)),
                ("k64_eq8_q8r24", dict(k=64, n_eq=8, m=96, n_quad=8, rank=24))):
    cs = workloads.build_constraints(workloads.corridor_like(seed=5, **kw))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    B = 262144
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    print(json.dumps({"set": tag, "n": cs.n, "k": cs.k, "kernel_family": dp.info().mfma_f32,
                      "fwd_ms": time_call(lambda: ops.project_raw(v, dp, want_active=False)),
                      "fwd_track_ms": time_call(lambda: ops.project_raw(v, dp, want_active=True))}))
