#!/usr/bin/env python
"""Developer measurement: the shader clock a forward kernel of config 3 actually runs at (s_memtime against the 100 MHz
s_memrealtime, entry to exit of one wave per 16 workgroups), with a -DRAYEN_WL_CLOCK / -DRAYEN_IO_CLOCK build:
    RAYEN_HIP_LIBRARY=.../librayen_mfma_pair_wl_clock.so python scripts/ubench/wl_clock.py --schedule 3 [--batches ...]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rayen_amd import _lib, ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--schedule", type=int, default=3)
ap.add_argument("--batches", default="262144,1048576")
ap.add_argument("--reserve", type=int, default=0)
args = ap.parse_args()
lib = _lib.load()
lib.rayen_pair_schedule(args.schedule)
lib.rayen_reserve_cus(args.reserve)
raw = ctypes.CDLL(os.environ["RAYEN_HIP_LIBRARY"])
fn = getattr(raw, "rayen_debug_wl_clock" if args.schedule == 3 else "rayen_debug_io_clock")
cs = workloads.build_constraints(workloads.make_raw("c3", seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
for B in [int(b) for b in args.batches.split(",")]:
    x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    y = torch.empty(B, cs.k, device="cuda")
    for _ in range(300):
        ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        ops.project_raw(x, dp, want_active=False, want_kappa=False, out=y)
    e1.record()
    torch.cuda.synchronize()
    buf = np.zeros(64, dtype=np.uint64)
    assert fn(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    b = buf.reshape(16, 4).astype(np.float64)
    b = b[b[:, 3] > b[:, 1]]        # (probes of workgroups that ran)
    ticks, real = b[:, 2] - b[:, 0], (b[:, 3] - b[:, 1]) / 100e6
    ghz = ticks / real / 1e9
    print(f"reserve {args.reserve} schedule {args.schedule} kernel {lib.rayen_last_forward_kernel()} B={B}: {e0.elapsed_time(e1) * 10:.1f} us per launch; "
          f"wave life {np.median(real) * 1e6:.1f} us, {np.median(ticks):.0f} shader clocks -> {np.median(ghz):.3f} GHz "
          f"(min {ghz.min():.3f}, max {ghz.max():.3f})", flush=True)
