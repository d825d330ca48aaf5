"""Config-3 backward alone (200 calls) for `rocprofv3 --kernel-trace --stats`: which kernels the time goes to."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
cs = workloads.build_constraints(workloads.make_raw(sys.argv[1] if len(sys.argv) > 1 else "c3", seed=0))
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
B = 262144
v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
_, kappa, active = ops.project_raw(v, dp, want_active=True)
for _ in range(200):
    ops.backward_raw(v, kappa, active, g, dp)
torch.cuda.synchronize()
