"""Feasibility with and without the inward bias (RAYEN_PREPARE_INWARD_BIAS / RAYEN_INWARD_BIAS, include/rayen_hip.h): for each
config and each eps, fp64 residuals of the fp32 outputs on 65 536 rows -- max, rows > 0, rows > 1e-6 -- and the distance of
the biased outputs from the unbiased ones (the parity cost: per-row inf-norm relative).  One JSON line per (config, eps).
    python scripts/ubench/inward_bias.py [c3 c5 c5r c2 ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from rayen_amd import workloads  # noqa: E402
from rayen_amd.constraint_module import ConstraintModule  # noqa: E402

EPS = (0.0, 2.0 ** -23, 2.0 ** -22, 2.0 ** -21, 2.0 ** -20, 2.0 ** -19)


def main(configs):
    dev = torch.device("cuda", 0)
    for name in configs:
        raw = workloads.make_raw(name, seed=0)
        cs = workloads.build_constraints(raw)
        rng = workloads.CONFIGS[name][3]
        B = min(65536, workloads.CONFIGS[name][2])
        x = torch.empty(B, cs.n, 1, device=dev).uniform_(-rng, rng, generator=torch.Generator(device=dev).manual_seed(1000))
        base = None
        for eps in EPS:
            os.environ["RAYEN_INWARD_BIAS"] = repr(eps)
            try:
                layer = ConstraintModule(cs, method="RAYEN", create_map=False).to(dev)
                y = layer(x)[:, :, 0].double().cpu().numpy()
            finally:
                del os.environ["RAYEN_INWARD_BIAS"]
            if base is None:
                base = y
            rows = cs.getViolationRows(y)
            shift = np.abs(y - base).max(axis=1) / np.maximum(np.abs(base).max(axis=1), 1e-30)
            print(json.dumps({"config": name, "eps": eps, "eps_log2": (None if eps == 0 else float(np.log2(eps))), "rows": B,
                              "max_violation": float(rows.max()), "violations_gt_0": int((rows > 0).sum()),
                              "violations_gt_1e-6": int((rows > 1e-6).sum()), "violations_gt_1e-7": int((rows > 1e-7).sum()),
                              "max_rel_shift_vs_unbiased": float(shift.max())}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:] or ["c3", "c5", "c5r", "c2", "c4"])
