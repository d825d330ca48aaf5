import os, sys, torch, json
sys.path.insert(0, os.getcwd())
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mult in (1, 2, 4):
    raw = workloads.random_lin_quad_soc(k=64, m=128 * mult, n_quad=4 * mult, n_soc=2 * mult, seed=0)
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    for B in (262144, 1048576):
        x = torch.empty(B, 64, device="cuda").uniform_(-1, 1)
        dp, _ = layer.device_pack(x.device)
        ms = t(lambda: ops.project_raw(x, dp, want_active=False))
        tiles = 1 + 4 * mult + 4 * mult * 1.5 + 2 * mult * 2
        # per SIMD: B/64 groups over 1024 SIMDs, tiles * 64 MFMA * 64 cycles each
        ideal_cyc = B / 64 / 1024 * tiles * 4096
        print(json.dumps({"mult": mult, "B": B, "ms": round(ms, 4), "tile_equiv": tiles,
                          "ideal_ms_at_2.3GHz": round(ideal_cyc / 2.3e9 * 1e3, 4),
                          "eff_vs_2.3GHz": round(ideal_cyc / 2.3e9 * 1e3 / ms, 3)}))
