#!/usr/bin/env python
"""Developer: forward / backward of LMI-only sets at the sweep's shapes (RAYEN_LB_GRID_MULT A/B)."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from rayen_amd import constraints, ops                    # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402
B = 2000
os.environ["RAYEN_LMI_BLOCK"] = "1"
out = {"nth": os.environ.get("RAYEN_LB_NTH", "-"), "512_upto": os.environ.get("RAYEN_LB_512_UPTO", "257"), "256_upto": os.environ.get("RAYEN_LB_256_UPTO", "-"), "dtype": os.environ.get("LMI_DTYPE", "f32")}
SHAPES = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(100, 100), (150, 100), (196, 100), (220, 100), (250, 100), (300, 100)]
for r_F, k in SHAPES:
    rng = np.random.default_rng(r_F * 7 + k)
    F = []
    for _ in range(k):
        tmp = rng.uniform(-1, 1, size=(r_F, r_F)); F.append((tmp + tmp.T) / 2)
    tmp = rng.uniform(-1, 1, size=(r_F, r_F)); F.append(tmp @ tmp.T + 0.5 * np.eye(r_F))
    dt = torch.float64 if os.environ.get("LMI_DTYPE") == "f64" else torch.float32
    torch.set_default_dtype(dt)
    cs = constraints.ConvexConstraints(lc=None, qcs=[], socs=[], lmic=constraints.LMIConstraint(F), y0=np.zeros((k, 1)))
    layer = ConstraintModule(cs, create_map=False).cuda()
    v = torch.empty(B, cs.n, device="cuda", dtype=dt).uniform_(-1, 1)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    y, kappa, active = ops.project_raw(v, dp, want_active=True)
    g = torch.ones(B, cs.k, device="cuda", dtype=dt)
    def t(fn, reps=4):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) / reps, 3)
    out[f"r{r_F}_k{k}"] = [t(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False)), t(lambda: ops.backward_raw(v, kappa, active, g, dp))]
print(json.dumps(out), flush=True)
