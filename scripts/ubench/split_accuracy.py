"""Row errors of the three fp32 forward families against the fp64 kernel on 100 000 directions per config
(DESIGN.md 4.0 / 4.0b).   python scripts/ubench/split_accuracy.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from helpers import rel_err_rows
from rayen_amd import workloads
from rayen_amd.constraint_module import ConstraintModule


def layer(cs, dtype, mode=None):
    prev = torch.get_default_dtype(); torch.set_default_dtype(dtype)
    if mode is not None:
        os.environ["RAYEN_FP32_MODE"] = mode
    try:
        m = ConstraintModule(cs, create_map=False).cuda()
        m.device_pack(torch.device("cuda", 0))
        return m
    finally:
        torch.set_default_dtype(prev)
        os.environ.pop("RAYEN_FP32_MODE", None)


for name in ("c2", "c3", "c5"):
    cs = workloads.build_constraints(workloads.make_raw(name, seed=13))
    lt = layer(cs, torch.float64)
    for scale in (1.5, 1e-3, 300.0):
        x = torch.empty(100000, cs.n, 1).uniform_(-scale, scale, generator=torch.Generator().manual_seed(15))
        yt = lt(x.double().cuda()).cpu().numpy()[:, :, 0]
        row = [f"{name} |v|<{scale:g}"]
        for tag, mode in (("f16 pairs", "3"), ("bf16 triples", "2"), ("exact fp32", "1")):
            y = layer(cs, torch.float32, mode)(x.cuda()).cpu().double().numpy()[:, :, 0]
            e = rel_err_rows(y, yt)
            row.append("%s max %.2e mean %.2e" % (tag, e.max(), e.mean()))
        print(" | ".join(row), flush=True)
