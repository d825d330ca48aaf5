"""Where do the W-stationary kernel's outputs differ from the plain pair kernel's?  (developer aid)"""
import sys
import numpy as np
import torch
from rayen_amd import _lib, ops, workloads
from rayen_amd.constraint_module import ConstraintModule

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
raw = workloads.make_raw("c3", seed=7) if name == "c3" else workloads.random_lin_quad_soc(k=64, m=300, n_quad=0, n_soc=0, seed=41)
cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, method="RAYEN", create_map=False).to("cuda")
dp, _ = layer.device_pack(torch.device("cuda", 0))
gen = torch.Generator(device="cuda").manual_seed(B)
v = torch.empty(B, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
lib = _lib.load()
lib.rayen_pair_schedule(2)
y1, k1, _ = ops.project_raw(v, dp, want_active=False)
f1 = lib.rayen_last_forward_kernel()
y1, k1 = y1.clone(), k1.clone()
lib.rayen_pair_schedule(0)
y2, k2, a2 = ops.project_raw(v, dp, want_active=True)
f2 = lib.rayen_last_forward_kernel()
print("families", f1, f2)
dk = (k1 != k2).cpu().numpy()
dy = (y1 != y2).any(dim=1).cpu().numpy()
print("kappa mismatches", dk.sum(), "of", B, "| y row mismatches", dy.sum())
idx = np.nonzero(dk)[0]
print("first mismatching rows", idx[:40])
print("row mod 64 histogram", np.bincount(idx % 64, minlength=64))
print("group index histogram (first 20 groups)", np.bincount(idx // 64)[:20])
seg = a2[:, 0].cpu().numpy()
print("active segment of mismatching rows (plain kernel)", np.bincount(seg[idx] + 1))
print("active segment overall", np.bincount(seg + 1))
rel = ((k1 - k2).abs() / k2.abs().clamp_min(1e-30)).cpu().numpy()
print("max rel diff kappa", rel.max(), "median of mismatches", np.median(rel[idx]) if len(idx) else 0)
for i in idx[:10]:
    print(i, float(k1[i]), float(k2[i]), int(seg[i]))
idy = np.nonzero(dy & ~dk)[0]
print("rows with equal kappa but different y:", len(idy), idy[:20])
if len(idy):
    i = idy[0]
    print(y1[i].cpu().numpy() - y2[i].cpu().numpy())
