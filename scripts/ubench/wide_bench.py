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
for k in (32, 64, 96, 128):
    raw = workloads.random_lin_quad_soc(k=k, m=2 * k, n_quad=4, n_soc=2, seed=1)
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    B = 262144
    x = torch.empty(B, k, device="cuda").uniform_(-1, 1)
    dp, _ = layer.device_pack(x.device)
    bytes_pp, flops_pp = workloads.algorithmic_work(cs)
    ms = t(lambda: ops.project_raw(x, dp, want_active=False))
    msg = t(lambda: ops.project_raw(x, dp, want_active=False, force_generic=True), 3)
    print(json.dumps({"k": k, "mfma": dp.info().mfma_f32, "ms": round(ms, 4), "Mproj_s": round(B / ms / 1e3, 1),
                      "TF_alg": round(flops_pp * B / ms / 1e9, 1), "generic_ms": round(msg, 3)}))
