#!/usr/bin/env python
"""Developer analysis of the headline kernel's walk (config 3, mfma_pair_io_kernel<2,false>) from s_memtime stamps:
    bash scripts/ubench/tu_variant.sh rayen_mfma_pair_io iostamps -DRAYEN_IO_STAMPS -fno-slp-vectorize
    RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_mfma_pair_io_iostamps.so python scripts/ubench/io_stamps.py
Per group of a wave (two at B = 262 144): entry -> first rows in, per tile [top -> burst + row operations issued -> epilogue
done -> next top], end of walk -> drain -> boundary code done; s_memtime ticks (100 MHz constant clock on gfx950: 10 ns)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, ops, workloads                   # noqa: E402
from rayen_amd.constraint_module import ConstraintModule     # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
raw = workloads.make_raw(name, seed=0)
cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
B = 262144
x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
y = torch.empty(B, cs.k, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(300):
    ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
torch.cuda.synchronize()
ev0.record()
for _ in range(100):
    ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
ev1.record()
torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) * 10.0
assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR_IO
lib = _lib.load()
buf = np.zeros(16 * 3 * 2 * 32 * 4, dtype=np.uint64)
lib.rayen_debug_io_stamps.restype = ctypes.c_int
lib.rayen_debug_io_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.rayen_debug_io_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(48, 2, 32, 4).astype(np.float64)
rows = [r for r in range(48) if st[r, 0, 31, 0] > 0 and st[r, 1, 31, 2] > 0]
n_items = int(max(np.max(np.nonzero(st[r, 1, :30, 0])[0]) for r in rows)) + 1
print(f"{name}: {len(rows)} stamped waves, {n_items} tiles per walk; kernel {us:.1f} us per launch (with the stamps); unit: s_memtime ticks")
entry = np.array([st[r, 0, 31, 3] for r in rows])
for rnd in (0, 1):
    top = np.array([st[r, rnd, 31, 0] for r in rows])
    t0 = np.array([st[r, rnd, :n_items, 0] for r in rows])
    t1 = np.array([st[r, rnd, :n_items, 1] for r in rows])
    t2 = np.array([st[r, rnd, :n_items, 2] for r in rows])
    wend = np.array([st[r, rnd, 31, 1] for r in rows])
    dend = np.array([st[r, rnd, 31, 2] for r in rows])
    bend = np.array([st[r, rnd, 30, 0] for r in rows])
    send = np.array([st[r, rnd, 30, 1] for r in rows])
    print(f"-- group {rnd}: " + (f"entry -> top (first rows requested and landed) {np.mean(top - entry):.0f}; " if rnd == 0 else "")
          + f"top -> first tile {np.mean(t0[:, 0] - top):.0f}; walk {np.mean(wend - t0[:, 0]):.0f} (per tile {np.mean(wend - t0[:, 0]) / n_items:.0f}); "
          f"drain {np.mean(dend - wend):.0f}; boundary code {np.mean(bend - dend):.0f}"
          + (f"; last rows out (burst store issued) {np.mean(send - bend):.0f}" if rnd == 1 else ""))
    print("   tile  burst+io  epilogue  to-next")
    nxt = np.concatenate((t0[:, 1:], wend[:, None]), axis=1)
    for it in range(n_items):
        print(f"   {it:4d} {np.mean(t1[:, it] - t0[:, it]):9.0f} {np.mean(t2[:, it] - t1[:, it]):9.0f} {np.mean(nxt[:, it] - t2[:, it]):8.0f}")
    print(f"   sum  {np.mean(t1 - t0, 0).sum():9.0f} {np.mean(t2 - t1, 0).sum():9.0f} {np.mean(nxt - t2, 0).sum():8.0f}")
# the two partners of a SIMD (waves 0 and 4 of a workgroup): phase of their bursts
print("-- partners (waves 0 and 4 of a workgroup share a SIMD): tile-top times of group 1 relative to wave 0's group top")
for blk in range(16):
    r0, r4 = blk * 3 + 0, blk * 3 + 2
    if r0 in rows and r4 in rows:
        base = st[r0, 1, 31, 0]
        a = " ".join(f"{st[r0, 1, it, 0] - base:6.0f}" for it in range(n_items))
        b = " ".join(f"{st[r4, 1, it, 0] - base:6.0f}" for it in range(n_items))
        print(f"   wg {blk * 64}: wave 0 tops {a}\n            wave 4 tops {b}")
        break
