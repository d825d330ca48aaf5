import os, sys, json, torch, numpy as np
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B = 262144
for name, raw in (("k60 eq10 packed", workloads.corridor_like(k=60, n_eq=10, m=160, n_quad=40, rank=3, seed=1)),
                  ("k64 eq16 packed", workloads.corridor_like(k=64, n_eq=16, m=128, n_quad=24, rank=4, seed=2))):
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    a = ops.backward_raw(v, kappa, active, g, dp); b = ops.backward_raw(v, kappa, active, g, dp, force_generic=True)
    err = float(((a - b).abs().amax(1) / b.abs().amax(1).clamp_min(1e-12)).median())
    print(json.dumps({"set": name, "n": cs.n, "k": cs.k, "bwd_ms": round(t(lambda: ops.backward_raw(v, kappa, active, g, dp)), 4),
                      "generic_ms": round(t(lambda: ops.backward_raw(v, kappa, active, g, dp, force_generic=True)), 4), "median_rel_diff": err}))
