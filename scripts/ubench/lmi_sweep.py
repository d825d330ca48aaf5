"""The reference's LMI timing sweep (examples/scripts/time_analysis.py:157-188: random symmetric F_i, r_F x r_F, k of them,
y0 = 0, 2000 samples) on the kernels of this build, sizes the LDS of one wave holds: which kernel served, forward time,
tracked forward + backward time, and the oracle (reference op sequence, CPU) on the same inputs for the smaller cases.
    python scripts/ubench/lmi_sweep.py [--oracle]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from rayen_amd import _lib, constraints, ops              # noqa: E402
from rayen_amd.constraint_module import ConstraintModule   # noqa: E402

NAMES = {1: "lane", 6: "lmi_quad", 7: "lmi_wave", 10: "lmi_block"}


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 2000
for r_F, k in ((10, 100), (50, 100), (100, 100), (100, 500), (100, 1000), (150, 100), (180, 100), (250, 100), (300, 100)):
    rng = np.random.default_rng(r_F * 7 + k)
    F = []
    for _ in range(k):
        tmp = rng.uniform(-1, 1, size=(r_F, r_F))
        F.append((tmp + tmp.T) / 2)
    tmp = rng.uniform(-1, 1, size=(r_F, r_F))
    F.append(tmp @ tmp.T + 0.5 * np.eye(r_F))
    t0 = time.time()
    cs = constraints.ConvexConstraints(lc=None, qcs=[], socs=[], lmic=constraints.LMIConstraint(F), y0=np.zeros((k, 1)))
    layer = ConstraintModule(cs, create_map=False).cuda()
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
    try:
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        setup = time.time() - t0
        y, kappa, active = ops.project_raw(v, dp, want_active=True)
    except _lib.RayenError as err:
        setup = time.time() - t0
        if err.code != _lib.E_UNSUPPORTED:
            raise
        # beyond one wave's LDS (r > ~190): no kernel holds the matrix; the module evaluates the packed form with the
        # device's libraries (rocBLAS GEMM + rocSOLVER eigvalsh through torch, rayen_amd/eager.py) and says so once
        import warnings
        layer.check_nan = False
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x3 = v.unsqueeze(2)
            y = layer(x3)[:, :, 0]
            kap = layer.computeKappa(x3)[:, 0, 0]

            def fwd_bwd():
                xg = x3.detach().requires_grad_(True)
                (layer(xg)[:, :, 0] * g).sum().backward()
            out = {"r_F": r_F, "k": k, "B": B, "kernel": "device libraries (rocBLAS + rocSOLVER through torch; no HIP kernel holds r > ~190)",
                   "setup_s": round(setup, 2), "fwd_ms": round(t(lambda: layer(x3), reps=3), 4),
                   "fwd_plus_bwd_ms": round(t(fwd_bwd, reps=3), 4),
                   "max_violation": float(cs.getMaxViolation(y[:256].cpu().double().numpy())),
                   "clipped": float((kap > 1).float().mean())}
        print(json.dumps(out), flush=True)
        continue
    fam = _lib.load().rayen_last_forward_kernel()
    out = {"r_F": r_F, "k": k, "B": B, "kernel": NAMES.get(fam, fam), "setup_s": round(setup, 2),
           "fwd_ms": round(t(lambda: ops.project_raw(v, dp)), 4),
           "bwd_ms": round(t(lambda: ops.backward_raw(v, kappa, active, g, dp)), 4),
           "max_violation": float(cs.getMaxViolation(y[:256].cpu().double().numpy())),
           "clipped": float((kappa > 1).float().mean())}
    if "--oracle" in sys.argv and r_F * r_F * k <= 100 * 100 * 500:
        from oracle import rayen_oracle as oracle
        sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
        from helpers import csd_from_cs
        buf = oracle.precompute(csd_from_cs(cs), torch.float32)
        x = v.cpu().unsqueeze(2)
        oracle.forward(buf, x[:64])
        t1 = time.time()
        y_ref = oracle.forward(buf, x)
        out["oracle_cpu_fwd_ms"] = round((time.time() - t1) * 1e3, 1)
        out["max_rel_diff_vs_oracle_fp32"] = float(((y.cpu() - y_ref[:, :, 0]).abs().amax(1) / y_ref[:, :, 0].abs().amax(1).clamp_min(1e-30)).max())
    print(json.dumps(out), flush=True)
