import os, sys, time, torch, json
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
for name in ("c1", "c2"):
    raw = workloads.make_raw(name, seed=0); cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda(); layer.check_nan = False
    B = workloads.CONFIGS[name][2]
    x = torch.empty(B, cs.n, 1, device="cuda").uniform_(-1, 1)
    with torch.no_grad():
        for _ in range(10): layer(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): layer(x)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / 200
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): layer(x)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            for _ in range(20): y = layer(x)
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / 400
    print(json.dumps({"config": name, "B": B, "eager_us_per_call": round(eager * 1e6, 2), "graph_us_per_call": round(graph * 1e6, 2),
                      "eager_Mproj_s": round(B / eager / 1e6, 1), "graph_Mproj_s": round(B / graph / 1e6, 1)}))
