#!/usr/bin/env python
"""Developer micro-benchmark: time the C-ABI entry points per config and per kernel path.

    python scripts/kernel_bench.py [--configs c1,c2,c3,c4,c5] [--reps 30]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from rayen_amd import ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402


def time_call(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2,c3,c4,c5")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=0)
    args = ap.parse_args()
    for name in args.configs.split(","):
        raw = workloads.make_raw(name, seed=0)
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, create_map=False).cuda()
        B = args.batch or min(workloads.CONFIGS[name][2], 262144)
        x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
        dp, _ = layer.device_pack(x.device)
        info = dp.info()
        bytes_pp, flops_pp = workloads.algorithmic_work(cs)
        row = {"config": name, "B": B, "k": cs.k, "n": cs.n, "rows": info.n_rows, "mfma": info.mfma_f32,
               "generic_block": info.generic_block}
        for label, force, track in (("auto", False, False), ("auto_track", False, True), ("generic", True, False)):
            t = time_call(lambda: ops.project_raw(x, dp, force_generic=force, want_active=track), args.reps)
            row[label] = {"ms": round(t * 1e3, 4), "Mproj_s": round(B / t / 1e6, 1),
                          "TFLOPs_alg": round(flops_pp * B / t / 1e12, 2),
                          "GBs_alg": round(bytes_pp * B / t / 1e9, 1)}
        x64 = x.double()
        layer64 = ConstraintModule(cs, create_map=False).double().cuda()
        dp64, _ = layer64.device_pack(x.device)
        t = time_call(lambda: ops.project_raw(x64, dp64), max(3, args.reps // 3))
        row["fp64"] = {"ms": round(t * 1e3, 4), "Mproj_s": round(B / t / 1e6, 1)}
        print(json.dumps(row))


if __name__ == "__main__":
    main()
