"""s_memtime stamps of workgroup 0 of the W-stationary kernel (a -DRAYEN_WS_ABL=256 build):
    RAYEN_HIP_LIBRARY=scripts/ubench/variants/librayen_mfma_pair_ws_stamps.so python scripts/ubench/ws_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from rayen_amd import _lib, ops, workloads
from rayen_amd.constraint_module import ConstraintModule

raw = workloads.make_raw("c3", seed=0)
cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
B = 262144
x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
_lib.load().rayen_pair_schedule(2)
for _ in range(30):
    y, kappa, _ = ops.project_raw(x, dp, want_active=False, want_kappa=True)
torch.cuda.synchronize()
assert _lib.load().rayen_last_forward_kernel() == _lib.KERNEL_PAIR_WS
st = kappa.view(torch.int32)[:4 * 2 * 16].cpu().numpy().astype(np.int64).reshape(4, 2, 16) & 0xFFFFFFFF
names = ["start", "stage0", "stage1", "stage2", "stage3", "stage4", "epilogue", "post"]
for w in range(4):
    for it in range(2):
        s = st[w, it]
        d = [(int(s[i + 1]) - int(s[i])) & 0xFFFFFFFF for i in range(7)]
        bar = (int(s[15]) - int(s[14])) & 0xFFFFFFFF
        print(f"wave {w} iteration {it + 2}: " + "  ".join(f"{n}={x}" for n, x in zip(names[1:], d)) + f"  | barrier wait {bar}  | total {(int(s[7]) - int(s[0])) & 0xFFFFFFFF}")
    print(f"wave {w}: iteration-to-iteration {(int(st[w, 1, 0]) - int(st[w, 0, 0])) & 0xFFFFFFFF}")
