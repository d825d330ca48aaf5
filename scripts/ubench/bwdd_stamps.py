#!/usr/bin/env python
"""Developer analysis of the dense-form f16-pair backward (config 3) from s_memtime stamps (shader clocks):
    bash scripts/ubench/tu_variant.sh rayen_mfma_bwdd stamps -DRAYEN_BWDD_STAMPS -fno-slp-vectorize
    RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_mfma_bwdd_stamps.so python scripts/ubench/bwdd_stamps.py [B]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads                         # noqa: E402
from rayen_amd.constraint_module import ConstraintModule     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
cs = workloads.build_constraints(workloads.make_raw("c3", seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
_, kappa, active = ops.project_raw(v, dp, want_active=True)
for _ in range(100):
    ops.backward_raw(v, kappa, active, g, dp)
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["RAYEN_HIP_LIBRARY"])
buf = np.zeros(64, dtype=np.uint64)
assert raw.rayen_debug_bwdd_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
st = buf.reshape(2, 32).astype(np.float64)
names = ["top -> rows landed, g.v", "split of v", "walk (12 tiles)", "closers (one pass)", "combine", "request next rows + stage + store"]
for w in (0, 1):
    life, real = st[w, 11] - st[w, 8], (st[w, 12] - st[w, 9]) / 100e6
    print(f"wave {4 * w}: life {life:.0f} clocks = {real * 1e6:.1f} us at {life / real / 1e9:.2f} GHz; entry -> barrier (forms, rows, aux in LDS) "
          f"{st[w, 10] - st[w, 8]:.0f}; groups walked {st[w, 13]:.0f}")
    print(f"wave {4 * w}: " + "; ".join(f"{names[i]} {st[w, i + 1] - st[w, i]:.0f}" for i in range(6)) + f"; total {st[w, 6] - st[w, 0]:.0f}")
