import os, sys, torch, json
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for name in ("c2", "c3", "c4", "c5"):
    for dtype in (torch.float32, torch.float64):
        torch.set_default_dtype(dtype)
        raw = workloads.make_raw(name, seed=0); cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, create_map=False).cuda(); layer.check_nan = False
        B = min(workloads.CONFIGS[name][2], 262144)
        x = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
        dp, pid = layer.device_pack(x.device)
        y, kappa, active = ops.project_raw(x, dp)
        g = torch.randn_like(y)
        fwd = t(lambda: ops.project_raw(x, dp))
        bwd = t(lambda: torch.ops.rayen_amd.ray_project_bwd(x, kappa, active, g, pid))
        print(json.dumps({"config": name, "dtype": str(dtype), "B": B, "fwd_ms": round(fwd, 4), "bwd_ms": round(bwd, 4),
                          "clipped_frac": round(float((kappa > 1).float().mean()), 3)}))
torch.set_default_dtype(torch.float32)
