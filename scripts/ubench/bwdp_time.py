"""Backward time of config 5 / 5r on the library RAYEN_HIP_LIBRARY points to (ablation builds of rayen_mfma_bwdp.hip)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
def t(fn, reps=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
out = []
for name in sys.argv[1:] or ["c5r", "c5"]:
    cs = workloads.build_constraints(workloads.make_raw(name, seed=0))
    dp, _ = ConstraintModule(cs, create_map=False).cuda().device_pack(torch.device("cuda", 0))
    B = 262144
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1); g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    gv = torch.empty_like(v)
    out.append("%s %.1f us" % (name, 1e3 * t(lambda: ops.backward_raw(v, kappa, active, g, dp))))
print(os.environ.get("RAYEN_HIP_LIBRARY", "default").split("/")[-1], " | ".join(out))
