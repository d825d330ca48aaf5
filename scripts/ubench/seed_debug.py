"""Forward error of one random set of tests/test_gpu_parity.py against the fp64 truth, per fp32 kernel family,
next to the yardsticks of _fp32_bound.  usage: python scripts/ubench/seed_debug.py SEED [SEED ...]"""
import importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
spec = importlib.util.spec_from_file_location("par", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
par = importlib.util.module_from_spec(spec); spec.loader.exec_module(par)
from helpers import csd_from_cs, rel_err_rows
from oracle import rayen_oracle as oracle
import packed_eval
from rayen_amd import ops, pack as _pack

for seed in map(int, sys.argv[1:]):
    raw = par._random_set(1000 + seed)
    rng = np.random.default_rng(seed)
    B = int(rng.choice([1, 31, 64, 65, 1000, 4099]))
    from rayen_amd import workloads
    from rayen_amd.constraint_module import ConstraintModule
    cs = workloads.build_constraints(raw)
    layer = ConstraintModule(cs, method="RAYEN", create_map=False)
    gen = torch.Generator().manual_seed(seed)
    x = torch.empty(B, cs.n, 1, dtype=torch.float32).uniform_(-2.0, 2.0, generator=gen)
    y_true = par._oracle_forward(cs, x.double(), torch.float64)
    y32 = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x).numpy()[:, :, 0]
    consts = layer.packed_constants()
    y_const, _, _ = packed_eval.evaluate(consts, x[:, :, 0].double().numpy())
    print(f"seed {seed}: k={cs.k} n={cs.n} m={cs.A_p.shape[0]} quad={len(cs.qcs)} soc={len(cs.socs)} "
          f"lmi={cs.lmic is not None} B={B}")
    print(f"  oracle fp32 err {rel_err_rows(y32, y_true).max():.3e}   constants rounding {rel_err_rows(y_const, y_true).max():.3e}")
    if not torch.cuda.is_available():
        continue
    for mode, tag in ((0, "default"), (1, "exact"), (2, "triple"), (3, "pair")):
        dp = _pack.DevicePack(consts, 0, fp32_mode=mode)
        info = dp.info()
        y = ops.project_raw(x[:, :, 0].cuda().contiguous(), dp)[0].cpu().numpy()
        e = rel_err_rows(y, y_true)
        print(f"  {tag:8s} family {info.mfma_f32} check pair/triple/exact {info.fp32_check_pair:.2e}/{info.fp32_check_split:.2e}/{info.fp32_check_exact:.2e} "
              f"err max {e.max():.3e} at {int(e.argmax())}  p99 {np.quantile(e, 0.99):.2e}")
        yg = ops.project_raw(x[:, :, 0].cuda().contiguous(), dp, force_generic=True)[0].cpu().numpy()
        print(f"           lane kernel err {rel_err_rows(yg, y_true).max():.3e}")
        dp.close()
