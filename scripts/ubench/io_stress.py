#!/usr/bin/env python
"""Stress of the counted waits of the trickled-row schedules (rayen_mfma_pair_io.hip): the same batch goes through the
trickled kernel and -- through a mis-aligned copy, which it declines -- through the plain pair kernel, REPS times with
fresh inputs and with other work on the chip in between; every output must agree bit for bit every time.
    python scripts/ubench/io_stress.py [--reps 300] [--configs c3,c5,c5r]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rayen_amd import _lib, ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--configs", default="c3,c5,c5r")
ap.add_argument("--batches", default="262144,393216,131072")
args = ap.parse_args()
lib = _lib.load()
for name in args.configs.split(","):
    extra = {"eq_n20": lambda: workloads.corridor_like(k=28, n_eq=8, m=330, n_quad=10, rank=3, seed=31),
             "id_n24": lambda: workloads.random_lin_quad_soc(k=24, m=260, n_quad=3, n_soc=1, seed=32),
             "id_n30_many": lambda: workloads.random_lin_quad_soc(k=30, m=300, n_quad=6, n_soc=2, seed=33),
             "n32": lambda: workloads.random_lin_quad_soc(k=32, m=300, n_quad=3, n_soc=2, seed=17)}
    cs = workloads.build_constraints(extra[name]() if name in extra else workloads.make_raw(name, seed=0))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    for B in [int(b) for b in args.batches.split(",")]:
        gen = torch.Generator(device="cuda").manual_seed(B)
        buf = torch.empty(B * cs.n + 4, device="cuda")
        noise = torch.empty(64 << 20, device="cuda")      # 256 MB: evicts L2 / MALL between the two runs
        bad, served = 0, 0
        for rep in range(args.reps):
            v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
            track = bool(rep & 1)
            y1, k1, a1 = ops.project_raw(v, dp, want_active=track)
            served += int(lib.rayen_last_forward_kernel() == _lib.KERNEL_PAIR_IO)
            if rep % 3 == 0:
                noise.add_(1.0)
            w = buf[1:1 + B * cs.n].view(B, cs.n)
            w.copy_(v)
            y2, k2, a2 = ops.project_raw(w, dp, want_active=track)
            assert lib.rayen_last_forward_kernel() == _lib.KERNEL_PAIR
            same = torch.equal(y1, y2) and (not track or (torch.equal(k1, k2) and torch.equal(a1, a2)))
            bad += int(not same)
            if not same and bad <= 4:
                rows = (y1 != y2).any(dim=1).nonzero().flatten()
                cols = (y1 != y2).any(dim=0).nonzero().flatten()
                nan1, nan2 = int(torch.isnan(y1).sum()), int(torch.isnan(y2).sum())
                diff = float((y1 - y2).abs().max())
                prev = lib.rayen_pair_schedule(0)                       # third opinion: the plain kernel on the ALIGNED rows
                y3, _, _ = ops.project_raw(v, dp, want_active=track)
                lib.rayen_pair_schedule(prev)
                y4, _, _ = ops.project_raw(v, dp, want_active=track)    # and the trickled kernel once more
                verdict = {"trickled_eq_third": bool(torch.equal(y1, y3)), "plain_misaligned_eq_third": bool(torch.equal(y2, y3)),
                           "trickled_again_eq_third": bool(torch.equal(y4, y3))}
                c0 = int(cols[0])
                stride_rows = 2048 * 64       # rows between two groups of one wave (2048 resident waves)
                prev_rows = rows - stride_rows
                if int(prev_rows.min()) >= 0:
                    verdict["bad_values_equal_previous_group_of_the_wave"] = bool(torch.equal(y1[rows, c0], y2[prev_rows, c0]))
                    verdict["bad_values_equal_group_before_that"] = bool(int(prev_rows.min()) >= stride_rows and torch.equal(y1[rows, c0], y2[prev_rows - stride_rows, c0]))
                verdict["bad"] = y1[rows, c0].tolist()[:4]
                verdict["good"] = y2[rows, c0].tolist()[:4]
                print(json.dumps(verdict), flush=True)
                print(json.dumps({"config": name, "B": B, "rep": rep, "track": track, "rows": int(rows.numel()),
                                  "first_rows": rows[:12].tolist(), "last_row": int(rows[-1]), "groups": sorted({int(r) // 64 for r in rows.tolist()})[:8],
                                  "cols": cols.tolist()[:48], "max_abs_diff": diff, "nan_trickled": nan1, "nan_plain": nan2}), flush=True)
        print(json.dumps({"config": name, "B": B, "reps": args.reps, "served_by_trickled_kernel": served, "mismatching_runs": bad}), flush=True)
