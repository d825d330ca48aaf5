#!/usr/bin/env python
"""Developer check behind the 1000-seed fuzz: on the fp32 LMI-backward seeds that exceed the 2e-4 bar, how far is the
forward's kappa from the fp64 oracle's, and how far is grad kappa . v from kappa (Euler's identity, exact for a
degree-1 candidate)?   python scripts/ubench/lmi_kappa_check.py 71 74 133"""
import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import test_gpu_backward as tb
from helpers import csd_from_cs
from oracle import rayen_oracle as oracle
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule
for seed in [int(a) for a in sys.argv[1:]]:
    cs = workloads.build_constraints(tb._random_lmi_set(seed))
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator().manual_seed(19)
    v = torch.empty(777, cs.n).uniform_(-2.0, 2.0, generator=gen); v[:20] *= 1e-3; v[20:22] = 0.0
    g = torch.empty(777, cs.k).uniform_(-1, 1, generator=gen)
    buf = oracle.precompute(csd_from_cs(cs), torch.float64)
    x = v.double().unsqueeze(2)
    k_true = oracle.compute_kappa(buf, x)[:, 0, 0].numpy()
    out = {}
    for dtype in (torch.float32, torch.float64):
        lay = ConstraintModule(cs, create_map=False).cuda().to(dtype)
        dpp, _ = lay.device_pack(torch.device("cuda", 0))
        vd, gd = v.to(dtype).cuda(), g.to(dtype).cuda()
        y, kappa, active = ops.project_raw(vd, dpp, want_active=True)
        gv = ops.backward_raw(vd, kappa, active, gd, dpp)
        kap = kappa.double().cpu().numpy()
        clipped = k_true > 1.0
        rel = np.abs(kap - k_true) / np.maximum(k_true, 1e-300)
        out[str(dtype)[6:]] = {"kappa_rel_err_max_on_clipped": float(rel[clipped].max()), "kappa_rel_err_median": float(np.median(rel[clipped])),
                               "argmax_row": int(np.flatnonzero(clipped)[rel[clipped].argmax()]), "forward_kernel": int(__import__("rayen_amd")._lib.load().rayen_last_forward_kernel()),
                               "max_abs_grad_on_clipped_rows": float(gv.double().abs().cpu().numpy()[clipped].max())}
    print(json.dumps({"seed": seed, "n": cs.n, "k": cs.k, "r": int(cs.lmic.all_F[0].shape[0]), **out}), flush=True)
