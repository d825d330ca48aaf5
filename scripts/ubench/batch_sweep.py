"""Kernel time of the fp32 forward against the batch size (multiples of one group per wave slot): separates the
fixed cost of a launch (start, first row load, last store) from the cost of one more group per wave.
    python scripts/ubench/batch_sweep.py [config] [fp32_mode]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
if len(sys.argv) > 2:
    os.environ["RAYEN_FP32_MODE"] = sys.argv[2]
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
cs = workloads.build_constraints(workloads.make_raw(cfg, seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
print(cfg, "family", dp.info().mfma_f32)
slot = 256 * 4 * 2 * 64
for mult in (0.25, 0.5, 1, 2, 3, 4, 6, 8, 16):
    B = int(slot * mult)
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    y = torch.empty(B, cs.k, device="cuda")
    for _ in range(200):
        ops.project_raw(v, dp, want_active=False, want_kappa=False, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        ops.project_raw(v, dp, want_active=False, want_kappa=False, out=y)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    print(json.dumps({"groups_per_wave": mult, "B": B, "ms": round(ms, 5), "Gproj_s": round(B / ms / 1e6, 3)}), flush=True)
