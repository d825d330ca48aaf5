"""Cost of the forward's optional outputs (kappa, the arg-max record) on config 3 / 5: same kernel family, same inputs."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule


def timed(fn, n=300):
    for _ in range(150):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for cfg in sys.argv[1:] or ["c3", "c5"]:
    cs = workloads.build_constraints(workloads.make_raw(cfg, seed=0))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    B = 262144
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    y = torch.empty(B, cs.k, device="cuda")
    row = {"config": cfg, "family": dp.info().mfma_f32}
    row["y only"] = timed(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False, out=y))
    row["y + kappa"] = timed(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=True, out=y))
    row["y + kappa + active"] = timed(lambda: ops.project_raw(v, dp, want_active=True, want_kappa=True, out=y))
    row["y + active"] = timed(lambda: ops.project_raw(v, dp, want_active=True, want_kappa=False, out=y))
    print(json.dumps({k: (round(x, 5) if isinstance(x, float) else x) for k, x in row.items()}), flush=True)
