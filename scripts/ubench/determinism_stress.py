#!/usr/bin/env python
"""Self-consistency of every kernel family under repetition: the same batch through the same kernel REPS times (other
work on the chip in between); every launch must return the bits of the first.  A hazard or a counted wait that is only
almost right shows up as a launch that differs.  Forward (with and without the arg-max record) and backward, fp32
families pinned by RAYEN_FP32_MODE / rayen_pair_schedule, fp64.
    python scripts/ubench/determinism_stress.py [--reps 200] [--configs c1,c2,c3,c4,c5,c5r]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rayen_amd import _lib, ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--configs", default="c1,c2,c3,c4,c5,c5r")
ap.add_argument("--modes", default="0,1,2")          # RAYEN_FP32_MODE: 0 default dispatch, 1 exact fp32, 2 bf16 triples
args = ap.parse_args()
lib = _lib.load()
noise = torch.empty(32 << 20, device="cuda")


def run(tag, fn, reps):
    first = fn()
    differing = 0
    for rep in range(reps):
        if rep % 4 == 0:
            noise.add_(1.0)
        out = fn()
        differing += int(not all(torch.equal(a, b) for a, b in zip(out, first) if a is not None))
    print(json.dumps(dict(tag, reps=reps, launches_differing_from_the_first=differing)), flush=True)


for name in args.configs.split(","):
    raw = workloads.make_raw(name, seed=0)
    cs = workloads.build_constraints(raw)
    B = workloads.CONFIGS[name][2]
    if name in ("c5", "c5r"):
        B = 393216            # three rounds: a wave's first, middle and last group
    for dtype in (torch.float32, torch.float64):
        for mode in (args.modes.split(",") if dtype == torch.float32 else ["0"]):
            os.environ["RAYEN_FP32_MODE"] = mode
            try:
                layer = ConstraintModule(cs, create_map=False).to("cuda").to(dtype)
                dp, _ = layer.device_pack(torch.device("cuda", 0))
            except Exception as e:       # a family that does not serve the set
                print(json.dumps({"config": name, "mode": mode, "skipped": str(e)[:80]}), flush=True)
                continue
            finally:
                os.environ.pop("RAYEN_FP32_MODE", None)
            gen = torch.Generator(device="cuda").manual_seed(5)
            v = torch.empty(B, cs.n, device="cuda", dtype=dtype).uniform_(-1.5, 1.5, generator=gen)
            g = torch.empty(B, cs.k, device="cuda", dtype=dtype).uniform_(-1, 1, generator=gen)
            schedules = (0, 1, 2) if (dtype == torch.float32 and mode == "0" and dp.info().mfma_f32 == 3) else (1,)
            for sched in schedules:
                prev = lib.rayen_pair_schedule(sched)
                for track in (False, True):
                    ops.project_raw(v, dp, want_active=track)
                    tag = {"config": name, "dtype": str(dtype)[6:], "fp32_mode": mode, "pair_schedule": sched, "B": B,
                           "what": "forward+record" if track else "forward", "kernel": int(lib.rayen_last_forward_kernel())}
                    run(tag, lambda: ops.project_raw(v, dp, want_active=track), args.reps)
                lib.rayen_pair_schedule(prev)
            _, kappa, active = ops.project_raw(v, dp, want_active=True)
            tag = {"config": name, "dtype": str(dtype)[6:], "fp32_mode": mode, "B": B, "what": "backward", "bwd_family": int(dp.info().bwd_f32)}
            run(tag, lambda: (ops.backward_raw(v, kappa, active, g, dp),), max(50, args.reps // 2))
