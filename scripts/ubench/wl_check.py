#!/usr/bin/env python
"""Developer check of the W-in-LDS schedule (rayen_mfma_pair_wl.hip) against the default schedule: same bits (y, kappa,
active) on several batch sizes incl. ragged ones, and the time of both (HIP events, after settling).
    python scripts/ubench/wl_check.py [--config c3] [--batches 262144,...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rayen_amd import _lib, ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--batches", default="262144,262143,200001,524288,1048576")
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--no-time", action="store_true")
args = ap.parse_args()
lib = _lib.load()
raw = workloads.make_raw(args.config, seed=0)
cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
torch.manual_seed(0)
for B in [int(b) for b in args.batches.split(",")]:
    x = torch.randn(B, cs.n, device="cuda") * torch.rand(B, 1, device="cuda") * 3
    res = {}
    for sched in (1, 3):
        lib.rayen_pair_schedule(sched)
        y = torch.full((B, cs.k), float("nan"), device="cuda")
        out = ops.project_raw(x, dp, want_active=True, want_kappa=True, out=y)
        torch.cuda.synchronize()
        res[sched] = (y.clone(), [o.clone() if torch.is_tensor(o) else o for o in (out if isinstance(out, (tuple, list)) else [out])],
                      lib.rayen_last_forward_kernel())
        y2 = torch.full((B, cs.k), float("nan"), device="cuda")
        ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y2)
        torch.cuda.synchronize()
        assert torch.equal(y2, y), f"schedule {sched}: tracked and plain instances differ"
    same_y = torch.equal(res[1][0], res[3][0])
    same_rest = all(torch.equal(a, b) for a, b in zip(res[1][1], res[3][1]) if torch.is_tensor(a))
    nd = int((res[1][0] != res[3][0]).any(dim=1).sum())
    print(f"B={B}: kernels {res[1][2]} / {res[3][2]}  y identical {same_y} ({nd} rows differ; max |d| "
          f"{float((res[1][0] - res[3][0]).abs().max()):.3e})  kappa/active identical {same_rest}  nan rows {int(res[3][0].isnan().any(dim=1).sum())}", flush=True)
    if args.no_time:
        continue
    t = {}
    for sched in (1, 3, 1, 3):
        lib.rayen_pair_schedule(sched)
        y = torch.empty(B, cs.k, device="cuda")
        call = lambda: ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)  # noqa: E731
        for _ in range(150):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        t.setdefault(sched, []).append(e0.elapsed_time(e1) / args.reps * 1e3)
    print(f"    time us: schedule 1 {t[1][0]:.1f} {t[1][1]:.1f} | schedule 3 {t[3][0]:.1f} {t[3][1]:.1f}", flush=True)
lib.rayen_pair_schedule(1)
