#!/usr/bin/env python
"""Developer: block against wave kernel around the dispatch threshold (forward / backward ms, B = 2000)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, constraints, ops                    # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402
B = 2000
NAMES = {1: "lane", 6: "quad", 7: "wave", 10: "block"}
for dt in (torch.float32, torch.float64):
    for r_F in (33, 36, 40, 50, 64, 80):
        k = 10
        rng = np.random.default_rng(r_F)
        F = []
        for _ in range(k):
            tmp = rng.uniform(-1, 1, size=(r_F, r_F)); F.append((tmp + tmp.T) / 2)
        tmp = rng.uniform(-1, 1, size=(r_F, r_F)); F.append(tmp @ tmp.T + 0.5 * np.eye(r_F))
        torch.set_default_dtype(dt)
        cs = constraints.ConvexConstraints(lc=None, qcs=[], socs=[], lmic=constraints.LMIConstraint(F), y0=np.zeros((k, 1)))
        row = {"dtype": str(dt)[6:], "r": r_F}
        for mode in ("1", "0"):
            os.environ["RAYEN_LMI_BLOCK"] = mode
            layer = ConstraintModule(cs, create_map=False).cuda()
            v = torch.empty(B, cs.n, device="cuda", dtype=dt).uniform_(-1, 1)
            dp, _ = layer.device_pack(torch.device("cuda", 0))
            y, kappa, active = ops.project_raw(v, dp, want_active=True)
            fam = NAMES.get(_lib.load().rayen_last_forward_kernel(), "?")
            g = torch.ones(B, cs.k, device="cuda", dtype=dt)
            def t(fn, reps=60):
                for _ in range(30): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps): fn()
                e1.record(); torch.cuda.synchronize()
                return round(e0.elapsed_time(e1) / reps, 4)
            row[f"pin{mode}"] = [fam, t(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False)), t(lambda: ops.backward_raw(v, kappa, active, g, dp))]
        print(json.dumps(row), flush=True)
