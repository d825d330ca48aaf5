#!/usr/bin/env python
"""Round 5: bisecting the intermittent y == y0 fault of the flat-row kernel's recording instance (DESIGN.md section 3,
"Repetition"): the PRE-FIX translation unit (git show 6851827^) with one experiment compiled in per library
(rayen_amd/csrc/variants/prefix, -DRAYEN_Y0_EXP=n), config 5 at the batch that failed, every launch with the arg-max
record, against the plain pair kernel on a mis-aligned copy of the same rows.
    RAYEN_HIP_LIBRARY=<variant .so> python scripts/ubench/y0_bisect.py [--reps 1500] [--batch 655360]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rayen_amd import _lib, ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=1500)
ap.add_argument("--batch", type=int, default=655360)
ap.add_argument("--config", default="c5")
args = ap.parse_args()
lib = _lib.load()
cs = workloads.build_constraints(workloads.make_raw(args.config, seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
B = args.batch
gen = torch.Generator(device="cuda").manual_seed(B)
buf = torch.empty(B * cs.n + 4, device="cuda")
noise = torch.empty(64 << 20, device="cuda")
bad, served, flagged, shown = 0, 0, 0, 0
cols_hit, lanes_hit = {}, {}
for rep in range(args.reps):
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    dp.nan_flag.zero_()
    y1, k1, a1 = ops.project_raw(v, dp, want_active=True)
    served += int(lib.rayen_last_forward_kernel() == _lib.KERNEL_PAIR_IO)
    flagged += int(dp.nan_flag.item() != 0)
    if rep % 3 == 0:
        noise.add_(1.0)
    w = buf[1:1 + B * cs.n].view(B, cs.n)
    w.copy_(v)
    y2, k2, a2 = ops.project_raw(w, dp, want_active=True)
    if torch.equal(y1, y2):
        continue
    bad += 1
    rows = (y1 != y2).any(dim=1).nonzero().flatten()
    cols = (y1 != y2).any(dim=0).nonzero().flatten()
    for c in cols.tolist():
        cols_hit[c] = cols_hit.get(c, 0) + 1
    key = (int(rows[0]) % 64, int(rows[-1]) % 64, int(rows.numel()))
    lanes_hit[str(key)] = lanes_hit.get(str(key), 0) + 1
    if shown < 3:
        shown += 1
        c0 = int(cols[0])
        y0c = float(layer.y0[c0, 0])
        print(json.dumps({"rep": rep, "rows": int(rows.numel()), "first_row_in_group": int(rows[0]) % 64, "group": int(rows[0]) // 64,
                          "round_of_wave": (int(rows[0]) // 64) // 2048, "cols": cols.tolist()[:8], "bad": y1[rows, c0].tolist()[:3],
                          "y0_of_col": y0c, "good": y2[rows, c0].tolist()[:3]}), flush=True)
print(json.dumps({"library": os.environ.get("RAYEN_HIP_LIBRARY", "default").split("/")[-1], "config": args.config, "B": B, "reps": args.reps,
                  "served_by_trickled_kernel": served, "mismatching_launches": bad, "nan_flag_launches": flagged,
                  "columns_hit": cols_hit, "row_pattern_hit(first%64,last%64,count)": lanes_hit}), flush=True)
