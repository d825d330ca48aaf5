#!/usr/bin/env python
"""Developer: a few forwards of one LMI-only set (for rocprofv3 counters):  python scripts/ubench/lmi_one.py <r> <k> [B] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import constraints, ops                    # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402

r_F, k = int(sys.argv[1]), int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.default_rng(r_F * 7 + k)
F = []
for _ in range(k):
    tmp = rng.uniform(-1, 1, size=(r_F, r_F))
    F.append((tmp + tmp.T) / 2)
tmp = rng.uniform(-1, 1, size=(r_F, r_F))
F.append(tmp @ tmp.T + 0.5 * np.eye(r_F))
cs = constraints.ConvexConstraints(lc=None, qcs=[], socs=[], lmic=constraints.LMIConstraint(F), y0=np.zeros((k, 1)))
layer = ConstraintModule(cs, create_map=False).cuda()
v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
dp, _ = layer.device_pack(torch.device("cuda", 0))
for _ in range(reps):
    ops.project_raw(v, dp, want_active=False, want_kappa=False)
torch.cuda.synchronize()
