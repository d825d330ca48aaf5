#!/usr/bin/env python
"""Runs every kernel family once per configuration (forward, forward+mapper, backward; fp32 and fp64) so that
one `rocprofv3 --kernel-trace --stats` pass lists them all:  scripts/profile_suite.sh <tag>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from rayen_amd import ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

REPS = 20
for dtype in (torch.float32, torch.float64):
    torch.set_default_dtype(dtype)
    for name, B in (("c1", 500), ("c2", 4096), ("c3", 262144), ("c4", 16384), ("c5", 262144)):
        cs = workloads.build_constraints(workloads.make_raw(name, seed=0))
        layer = ConstraintModule(cs, create_map=False).cuda()
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
        g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
        for _ in range(REPS):
            _, kappa, active = ops.project_raw(v, dp, want_active=True)
            ops.project_raw(v, dp, want_active=False)
            ops.backward_raw(v, kappa, active, g, dp)
        if dtype == torch.float32 and name == "c3":
            mapped = ConstraintModule(cs, input_dim=64, create_map=True).cuda()
            x = torch.empty(B, 64, device="cuda").uniform_(-1, 1)
            with torch.no_grad():
                for _ in range(REPS):
                    mapped(x)
torch.cuda.synchronize()
print("done")
