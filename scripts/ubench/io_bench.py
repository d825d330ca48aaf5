#!/usr/bin/env python
"""Developer timing of the f16-pair forward on config 3 over batch sizes (HIP events, 100 launches after settling):
    [RAYEN_HIP_LIBRARY=...] [RAYEN_PAIR_IO=0] python scripts/ubench/io_bench.py [--config c3] [--batches 131072,262144,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rayen_amd import _lib, ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--batches", default="131072,262144,524288,1048576")
ap.add_argument("--schedule", type=int, default=-1)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--track", action="store_true")
args = ap.parse_args()
if args.schedule >= 0:
    _lib.load().rayen_pair_schedule(args.schedule)
raw = workloads.make_raw(args.config, seed=0)
cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
out = []
for B in [int(b) for b in args.batches.split(",")]:
    x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    y = torch.empty(B, cs.k, device="cuda")
    call = lambda: ops.project_raw(x, dp, want_active=args.track, want_kappa=args.track, out=y)  # noqa: E731
    for _ in range(150):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    out.append(f"B={B}: {e0.elapsed_time(e1) / args.reps * 1e3:.1f} us")
print(os.environ.get("RAYEN_HIP_LIBRARY", "default").split("/")[-1], "io=" + os.environ.get("RAYEN_PAIR_IO", "1"),
      "kernel", _lib.load().rayen_last_forward_kernel(), "|", "  ".join(out))
