#!/usr/bin/env python
"""Developer micro-benchmark: forward (with active tracking) and both fp32 backward kernels."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from rayen_amd import ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402


def time_call(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


DTYPE = torch.float64 if os.environ.get("BWD_FP64") else torch.float32
torch.set_default_dtype(DTYPE)
for name in sys.argv[1:] or ["c2", "c3", "c5"]:
    cs = workloads.build_constraints(workloads.make_raw(name, seed=0))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    B = int(os.environ.get("BWD_B", "262144"))
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    row = {"config": name, "B": B, "clipped_frac": float((kappa > 1).float().mean()),
           "fwd_track_ms": time_call(lambda: ops.project_raw(v, dp, want_active=True)),
           "bwd_ms": time_call(lambda: ops.backward_raw(v, kappa, active, g, dp)),
           "dtype": str(DTYPE)}
    if DTYPE == torch.float32 or os.environ.get("BWD_GENERIC"):
        row["bwd_generic_ms"] = time_call(lambda: ops.backward_raw(v, kappa, active, g, dp, force_generic=True))
    print(json.dumps(row))
