"""Where the host time of one eager forward call goes (cProfile, config 1, B=500)."""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.getcwd())
from rayen_amd import workloads
from rayen_amd.constraint_module import ConstraintModule
raw = workloads.make_raw("c1", seed=0); cs = workloads.build_constraints(raw)
layer = ConstraintModule(cs, create_map=False).cuda(); layer.check_nan = False
x = torch.empty(500, cs.n, 1, device="cuda").uniform_(-1, 1)
with torch.no_grad():
    for _ in range(50): layer(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000): layer(x)
    torch.cuda.synchronize()
    print("eager us/call", (time.perf_counter() - t0) / 2000 * 1e6)
    pr = cProfile.Profile(); pr.enable()
    for _ in range(2000): layer(x)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
