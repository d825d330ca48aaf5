"""Training-step timings of the equality / packed-quadratic sets (configs 5, 5r): tracked forward, backward with and
without the bucketing workspace, the lane-per-sample backward; who is active (interior / a linear row / a limit).
    python scripts/ubench/bwd_c5.py [c5 c5r ...] [--B 262144] [--scale 1.0]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads                      # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402


def t(fn, reps=30):
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


ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="*", default=["c5", "c5r"])
ap.add_argument("--B", type=int, default=262144)
ap.add_argument("--scale", type=float, default=1.0)
args = ap.parse_args()
for name in args.configs:
    cs = workloads.build_constraints(workloads.make_raw(name, seed=0))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    v = torch.empty(args.B, cs.n, device="cuda").uniform_(-args.scale, args.scale)
    g = torch.empty(args.B, cs.k, device="cuda").uniform_(-1, 1)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    clipped = kappa > 1
    n_lin = int((clipped & (active[:, 0] == 0)).sum()) if active.ndim == 2 else -1
    a = ops.backward_raw(v, kappa, active, g, dp)
    b = ops.backward_raw(v, kappa, active, g, dp, force_generic=True)
    c = ops.backward_raw(v, kappa, active, g, dp, bucketed=False)
    den = b.abs().amax(1).clamp_min(1e-12)
    out = {"set": name, "n": cs.n, "k": cs.k, "B": args.B,
           "clipped": float(clipped.float().mean()), "clipped_on_segment_0": n_lin / args.B,
           "fwd_tracked_ms": round(t(lambda: ops.project_raw(v, dp, want_active=True)), 4),
           "fwd_ms": round(t(lambda: ops.project_raw(v, dp, want_active=False)), 4),
           "bwd_ms": round(t(lambda: ops.backward_raw(v, kappa, active, g, dp)), 4),
           "bwd_no_workspace_ms": round(t(lambda: ops.backward_raw(v, kappa, active, g, dp, bucketed=False)), 4),
           "bwd_lane_ms": round(t(lambda: ops.backward_raw(v, kappa, active, g, dp, force_generic=True)), 4),
           "max_rel_diff_vs_lane": float(((a - b).abs().amax(1) / den).max()),
           "max_rel_diff_plain_vs_lane": float(((c - b).abs().amax(1) / den).max())}
    print(json.dumps(out), flush=True)
