#!/usr/bin/env python
"""Developer A/B of the workgroup-per-sample LMI kernel's launch shape (the library is chosen with RAYEN_HIP_LIBRARY, the
512-thread range with RAYEN_LB_512_UPTO):  python scripts/ubench/lmi_block_ab.py"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import constraints, ops                    # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402

os.environ["RAYEN_LMI_BLOCK"] = "1"
B = 2000
out = {"lib": os.path.basename(os.environ.get("RAYEN_HIP_LIBRARY", "default")), "512_upto": os.environ.get("RAYEN_LB_512_UPTO", "128")}
for dtype in (torch.float32, torch.float64):
    for r_F in ((70, 100, 128, 150, 180, 196, 250, 280, 300) if dtype == torch.float32 else (100, 150, 196)):
        k = 10
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
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        fn = lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False)   # noqa: E731
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[f"{'f32' if dtype == torch.float32 else 'f64'}_r{r_F}"] = round(e0.elapsed_time(e1) / 4, 3)
print(json.dumps(out), flush=True)
