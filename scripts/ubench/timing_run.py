import os, sys, torch
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
raw = workloads.make_raw("c3", seed=0); cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
dp, _ = layer.device_pack(x.device)
dp.nan_flag = torch.zeros(512, dtype=torch.int32, device="cuda")
for _ in range(3): ops.project_raw(x, dp)
torch.cuda.synchronize()
dp.nan_flag.zero_()
ops.project_raw(x, dp); torch.cuda.synchronize()
ts = dp.nan_flag.cpu().numpy()[16:]
ts = ts[ts != 0]
d = (ts[1:].astype("int64") - ts[:-1].astype("int64")) & 0xffffffff
print("n stamps", len(ts)); print(d.tolist())
