#!/usr/bin/env python
"""Developer: where the workgroup-per-sample LMI kernel's cycles go.  Needs the profiling variant of the library:
    scripts/ubench/tu_variant.sh rayen_lmi_block prof -DRAYEN_LB_PROFILE
    RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_lmi_block_prof.so python scripts/ubench/lmi_block_prof.py
Prints, per shape, the cycles (s_memtime) workgroup 0 spent per sample between the barriers of the reduction."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, constraints, ops              # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402

lib = _lib.load()
prof = lib.rayen_debug_lb_prof
prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
prof.restype = None
SLOTS = ["column+sigma", "matvec", "w", "update", "S(v)", "sturm", "all", "samples"]

os.environ["RAYEN_LMI_BLOCK"] = "1"
B = 2000
for dtype in (torch.float32,):
    for r_F, k in ((100, 10), (150, 10), (250, 10), (280, 10), (250, 100)):
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
        ops.project_raw(v, dp, want_active=False)
        buf = (ctypes.c_ulonglong * 8)()
        prof(buf)                                           # (drop the warm-up's counts)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.project_raw(v, dp, want_active=False, want_kappa=False)
        e1.record()
        torch.cuda.synchronize()
        prof(buf)
        n = max(int(buf[7]), 1)
        row = {"r": r_F, "k": k, "ms": round(e0.elapsed_time(e1), 3), "samples_of_wg0": n}
        for name, val in zip(SLOTS[:7], list(buf)[:7]):
            row[name] = int(val) // n
        print(json.dumps(row), flush=True)
