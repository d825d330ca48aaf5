#!/usr/bin/env python
"""Developer helper: where the fp32 matrix-core backward's time goes -- config 3 against the same shape without its
quadratics and cones (no tile walk: what is left is the row I/O and the bookkeeping)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from rayen_amd import ops, workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402


def timeit(fn, reps=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


B = 262144
for name, raw in (("c3", workloads.make_raw("c3", seed=0)),
                  ("c3_lin_only", workloads.random_lin_quad_soc(k=64, m=128, n_quad=0, n_soc=0, seed=0)),
                  ("c3_1quad", workloads.random_lin_quad_soc(k=64, m=128, n_quad=1, n_soc=0, seed=0)),
                  ("c3_2quad", workloads.random_lin_quad_soc(k=64, m=128, n_quad=2, n_soc=0, seed=0))):
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, create_map=False).cuda()
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    v = torch.empty(B, cs.n, device="cuda").uniform_(-1, 1)
    g = torch.empty(B, cs.k, device="cuda").uniform_(-1, 1)
    _, kappa, active = ops.project_raw(v, dp, want_active=True)
    seg = active[:, 0]
    row = {"set": name, "clipped": float((kappa > 1).float().mean()),
           "active_hist": torch.bincount(seg + 1, minlength=len(dp.consts.segments) + 1).tolist(),
           "fwd_ms": timeit(lambda: ops.project_raw(v, dp, want_active=False, want_kappa=False)),
           "fwd_track_ms": timeit(lambda: ops.project_raw(v, dp, want_active=True)),
           "bwd_ms": timeit(lambda: ops.backward_raw(v, kappa, active, g, dp))}
    # the same backward on inputs sorted by active segment (what a bucketed walk would see, without skipping anything)
    order = torch.argsort(seg, stable=True)
    vs, ks, as_, gs = v[order].contiguous(), kappa[order].contiguous(), active[order].contiguous(), g[order].contiguous()
    row["bwd_sorted_inputs_ms"] = timeit(lambda: ops.backward_raw(vs, ks, as_, gs, dp))
    print(json.dumps(row), flush=True)
