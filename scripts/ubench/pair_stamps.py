"""Timestamps of three waves of the f16-pair kernel: build variant 320 (per tile) or 64 (per phase) with
scripts/ubench/pair_variants.py and point RAYEN_HIP_LIBRARY below at it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["RAYEN_HIP_LIBRARY"] = os.path.join(ROOT, "rayen_amd", "csrc", "variants", "librayen_hip_v320.so")
os.environ["RAYEN_FP32_MODE"] = "3"
import torch
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
cs = workloads.build_constraints(workloads.make_raw(cfg, seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
names = ["entry", "barrier", "g0 rows", "g0 split", "g0 walk", "g0 drain", "g0 stored", "g1 rows", "g1 split", "g1 walk", "g1 drain", "g1 stored", "end"]
for B in (32768, 262144, 524288):
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    for _ in range(50):
        y, kappa, _ = ops.project_raw(v, dp, want_active=False, want_kappa=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y, kappa, _ = ops.project_raw(v, dp, want_active=False, want_kappa=True); e1.record(); torch.cuda.synchronize()
    k = kappa.cpu()
    print("B", B, "kernel+launch us", round(e0.elapsed_time(e1) * 1e3, 1))
    for base, tag in ((0, "wave 0"), (32, "wave 4"), (64, "wave 1000")):
        vals = [int(x) for x in k[base:base + 24].tolist()]
        print("  ", tag, vals)
