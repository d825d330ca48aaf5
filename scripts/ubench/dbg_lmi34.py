import os, sys, importlib.util
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
spec = importlib.util.spec_from_file_location("tb", "tests/test_gpu_backward.py"); tb = importlib.util.module_from_spec(spec); spec.loader.exec_module(tb)
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 34
raw = tb._random_lmi_set(seed)
cs = workloads.build_constraints(raw)
print("k", cs.k, "n", cs.n, "r", cs.lmic.all_F[0].shape[0], "m", 0 if raw["A1"] is None else raw["A1"].shape[0])
layer = ConstraintModule(cs, create_map=False).cuda()
dp, _ = layer.device_pack(torch.device("cuda", 0))
gen = torch.Generator().manual_seed(19)
B = 777
v = torch.empty(B, cs.n).uniform_(-2.0, 2.0, generator=gen); v[:20] *= 1e-3; v[20:22] = 0.0
g = torch.empty(B, cs.k).uniform_(-1, 1, generator=gen)
vd, gd = v.cuda(), g.cuda()
y, kappa, active = ops.project_raw(vd, dp, want_active=True)
got = ops.backward_raw(vd, kappa, active, gd, dp).cpu().double()
lane = ops.backward_raw(vd, kappa, active, gd, dp, force_generic=True).cpu().double()
print("active segs", torch.unique(active[:, 0].cpu(), return_counts=True))
err = (got - lane).abs().amax(1) / lane.abs().amax(1).clamp_min(1e-12)
bad = torch.nonzero(err > 5e-3)[:, 0]
print("bad", len(bad), "of", B)
for i in bad[:8].tolist():
    print(i, "v", v[i].tolist(), "kappa", float(kappa[i]), "act", active[i].tolist(), "got", got[i].tolist(), "lane", lane[i].tolist())
