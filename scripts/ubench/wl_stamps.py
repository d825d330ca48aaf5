#!/usr/bin/env python
"""Developer analysis of the W-in-LDS walk (config 3) from s_memtime stamps (shader clocks):
    bash scripts/ubench/tu_variant.sh rayen_mfma_pair_wl stamps -DRAYEN_WL_STAMPS -fno-slp-vectorize
    RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_mfma_pair_wl_stamps.so python scripts/ubench/wl_stamps.py [B]
Second group of waves 0 and 4 of workgroup 0 (partners on one SIMD): per item [top -> burst + next reads issued -> epilogue done]."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, ops, workloads                   # noqa: E402
from rayen_amd.constraint_module import ConstraintModule     # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
cs = workloads.build_constraints(workloads.make_raw("c3", seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
lib = _lib.load()
lib.rayen_pair_schedule(3)
x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
y = torch.empty(B, cs.k, device="cuda")
for _ in range(300):
    ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
torch.cuda.synchronize()
raw = ctypes.CDLL(os.environ["RAYEN_HIP_LIBRARY"])
buf = np.zeros(2 * 40 * 4, dtype=np.uint64)
assert raw.rayen_debug_wl_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
st = buf.reshape(2, 40, 4).astype(np.float64)
n_items = int(np.max(np.nonzero(st[0, :38, 0])[0])) + 1
print(f"B={B}, kernel {lib.rayen_last_forward_kernel()}, {n_items} items; shader clocks")
for w in (0, 1):
    t0, t1, t2 = st[w, :n_items, 0], st[w, :n_items, 1], st[w, :n_items, 2]
    top, wend, gend = st[w, 38, 0], st[w, 38, 1], st[w, 38, 2]
    print(f"-- wave {4 * w}: group top -> first item {t0[0] - top:.0f} (split + request); walk {wend - t0[0]:.0f} "
          f"({(wend - t0[0]) / n_items:.0f} per item); kappa + next maxima + rows of y out {gend - wend:.0f}")
    print("   item  burst+reads  epilogue  to-next")
    for i in range(n_items):
        nxt = (t0[i + 1] if i + 1 < n_items else wend) - t2[i]
        print(f"   {i:4d}  {t1[i] - t0[i]:11.0f}  {t2[i] - t1[i]:8.0f}  {nxt:7.0f}")
    print(f"    sum  {np.sum(t1 - t0):11.0f}  {np.sum(t2 - t1):8.0f}")
print("   item tops of wave 4 relative to wave 0's first item:", " ".join(f"{v - st[0, 0, 0]:.0f}" for v in st[1, :n_items, 0]))
print("   item tops of wave 0:", " ".join(f"{v - st[0, 0, 0]:.0f}" for v in st[0, :n_items, 0]))
