import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
from helpers import rel_err_rows
from rayen_amd import workloads
from rayen_amd.constraint_module import ConstraintModule
def layer(cs, dtype):
    prev = torch.get_default_dtype(); torch.set_default_dtype(dtype)
    try: return ConstraintModule(cs, create_map=False).cuda()
    finally: torch.set_default_dtype(prev)
for name in ("c2", "c3", "c5"):
    cs = workloads.build_constraints(workloads.make_raw(name, seed=13))
    ls = layer(cs, torch.float32)
    os.environ["RAYEN_SPLIT_BF16"] = "0"; le = layer(cs, torch.float32); le.device_pack(torch.device("cuda", 0)); del os.environ["RAYEN_SPLIT_BF16"]
    lt = layer(cs, torch.float64)
    x = torch.empty(100000, cs.n, 1).uniform_(-1.5, 1.5, generator=torch.Generator().manual_seed(15))
    ys = ls(x.cuda()).cpu().double().numpy()[:, :, 0]; ye = le(x.cuda()).cpu().double().numpy()[:, :, 0]
    yt = lt(x.double().cuda()).cpu().numpy()[:, :, 0]
    es, ee = rel_err_rows(ys, yt), rel_err_rows(ye, yt)
    print(name, "split max %.2e mean %.2e | exact-fp32 max %.2e mean %.2e" % (es.max(), es.mean(), ee.max(), ee.mean()))
