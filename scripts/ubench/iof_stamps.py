#!/usr/bin/env python
"""Developer analysis of the flat-row kernel's tile walk (config 5) from s_memtime stamps:
    bash scripts/ubench/tu_variant.sh rayen_mfma_pair_io stamps -DRAYEN_IOF_STAMPS
    RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_mfma_pair_io_stamps.so python scripts/ubench/iof_stamps.py [config]
Per tile of a wave's second group: cycles from the tile's top to the point where its MFMA burst, A reloads and row
operations are issued, from there to the end of its epilogue, and to the next tile's top; and the group boundary."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, ops, workloads                   # noqa: E402
from rayen_amd.constraint_module import ConstraintModule     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c5"
raw = workloads.make_raw(name, seed=0)
cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
B = 262144
x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
y = torch.empty(B, cs.k, device="cuda")
for _ in range(200):
    ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
torch.cuda.synchronize()
assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR_IO
lib = _lib.load()
buf = np.zeros(16 * 3 * 256 * 4, dtype=np.uint64)
lib.rayen_debug_iof_stamps.restype = ctypes.c_int
lib.rayen_debug_iof_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.rayen_debug_iof_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(48, 256, 4).astype(np.float64)
rows = [r for r in range(48) if st[r, 0, 0] > 0 and st[r, 255, 2] > 0]
n_items = int(max(np.max(np.nonzero(st[r, :255, 0])[0]) for r in rows)) + 1
print(f"{name}: {len(rows)} stamped waves, {n_items} tiles per walk")
burst = np.array([st[r, :n_items, 1] - st[r, :n_items, 0] for r in rows])
epi = np.array([st[r, :n_items, 2] - st[r, :n_items, 1] for r in rows])
gap = np.array([np.append(st[r, 1:n_items, 0], st[r, 255, 1]) - st[r, :n_items, 2] for r in rows])
walk = np.array([st[r, 255, 1] - st[r, 255, 0] for r in rows])
drain = np.array([st[r, 255, 2] - st[r, 255, 1] for r in rows])
head = np.array([st[r, 0, 0] - st[r, 255, 0] for r in rows])
print(f"group: top -> first tile {head.mean():.0f}, walk {walk.mean():.0f} (per tile {walk.mean() / n_items:.0f}), drain {drain.mean():.0f}")
print("tile  burst+io  epilogue  to-next   (mean over the stamped waves, ticks ~ shader cycles)")
for it in range(n_items):
    print(f"{it:4d} {burst[:, it].mean():9.0f} {epi[:, it].mean():9.0f} {gap[:, it].mean():8.0f}")
print(f"sum  {burst.mean(0).sum():9.0f} {epi.mean(0).sum():9.0f} {gap.mean(0).sum():8.0f}")
desc = np.array([st[r, :n_items, 3] - st[r, :n_items, 1] for r in rows])
packed = [it for it in range(n_items) if (st[rows[0], it, 3] > 0)]
if packed:
    print("packed tiles: cycles from the end of the burst until the tile's descriptor (dependent scalar load) is there")
    print("  " + "  ".join(f"{it}:{desc[:, it].mean():.0f}" for it in packed))
