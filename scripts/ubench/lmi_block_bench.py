#!/usr/bin/env python
"""Round 5: the workgroup-per-sample LMI forward (rayen_lmi_block.h) against the wave-per-sample kernel (rayen_lmi_wave.h)
on the reference's sweep shapes (examples/scripts/time_analysis.py:157-188: random symmetric F_i, y0 = 0, 2000 samples).
    python scripts/ubench/lmi_block_bench.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, constraints, ops              # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402

NAMES = {1: "lane", 6: "lmi_quad", 7: "lmi_wave", 10: "lmi_block"}


def t(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 2000
for dtype in (torch.float32, torch.float64):
    for r_F, k in ((40, 10), (64, 10), (100, 10), (100, 100), (150, 10), (180, 10), (196, 10), (250, 10), (280, 10), (300, 10), (304, 10)):
        rng = np.random.default_rng(r_F * 7 + k)
        F = []
        for _ in range(k):
            tmp = rng.uniform(-1, 1, size=(r_F, r_F))
            F.append((tmp + tmp.T) / 2)
        tmp = rng.uniform(-1, 1, size=(r_F, r_F))
        F.append(tmp @ tmp.T + 0.5 * np.eye(r_F))
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            cs = constraints.ConvexConstraints(lc=None, qcs=[], socs=[], lmic=constraints.LMIConstraint(F), y0=np.zeros((k, 1)))
            layer = ConstraintModule(cs, create_map=False).cuda()
        finally:
            torch.set_default_dtype(prev)
        v = torch.empty(B, cs.n, device="cuda", dtype=dtype).uniform_(-1, 1)
        row = {"dtype": str(dtype).split(".")[-1], "r": r_F, "k": k, "B": B}
        ys = {}
        for mode in ("1", "0"):
            os.environ["RAYEN_LMI_BLOCK"] = mode
            try:
                dp, _ = layer.device_pack(torch.device("cuda", 0))
                y, kappa, _ = ops.project_raw(v, dp, want_active=False)
                fam = NAMES.get(_lib.load().rayen_last_forward_kernel(), "?")
                ms = t(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False))
                row[f"block={mode}"] = f"{fam} {ms:.3f} ms"
                ys[mode] = (y, kappa)
            except _lib.RayenError as err:
                row[f"block={mode}"] = f"refused ({err.code})"
        if len(ys) == 2:
            d = (ys["1"][1] - ys["0"][1]).abs() / ys["0"][1].abs().clamp_min(1e-30)
            row["max_rel_kappa_diff_block_vs_wave"] = float(d.max())
        print(json.dumps(row), flush=True)
