import os, sys, importlib.util
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, torch
spec = importlib.util.spec_from_file_location("tp", "tests/test_gpu_parity.py"); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from helpers import rel_err_rows
from rayen_amd import ops
import packed_eval
seed = int(sys.argv[1])
raw = tp._random_set(1000 + seed)
rng = np.random.default_rng(seed); B = int(rng.choice([1, 31, 64, 65, 1000, 4099]))
print("k", raw["A1"].shape if raw["A1"] is not None else None, "nP", len(raw["P"]), "nM", len(raw["M"]), "F", len(raw["F"]), "A2", None if raw["A2"] is None else raw["A2"].shape, "B", B)
gen = torch.Generator().manual_seed(seed)
def run(env):
    if env is None: os.environ.pop("RAYEN_SPLIT_BF16", None)
    else: os.environ["RAYEN_SPLIT_BF16"] = env
    cs, layer = tp._layer(raw, torch.float32)
    return cs, layer
cs, layer = run(None)
x = torch.empty(B, cs.n, 1, dtype=torch.float32).uniform_(-2.0, 2.0, generator=gen)
y_true = tp._oracle_forward(cs, x.double(), torch.float64)
y_ref = tp._oracle_forward(cs, x, torch.float32)
print("n", cs.n, "k", cs.k, "family", layer.device_pack(torch.device("cuda", 0))[0].info().mfma_f32)
y = layer(x.cuda()).cpu().numpy()[:, :, 0]
cs0, layer0 = run("0")
y0 = layer0(x.cuda()).cpu().numpy()[:, :, 0]
dp, _ = layer0.device_pack(torch.device("cuda", 0))
yg, _, _ = ops.project_raw(x[:, :, 0].cuda(), dp, force_generic=True)
y_const, _, _ = packed_eval.evaluate(layer.packed_constants(), x[:, :, 0].double().numpy())
for name, arr in (("split/default", y), ("exact fp32 family", y0), ("lane kernel", yg.cpu().numpy()), ("reference fp32", y_ref), ("constants in fp64", y_const)):
    e = rel_err_rows(arr, y_true)
    print("%-20s max %.3e at %d  mean %.3e" % (name, e.max(), int(e.argmax()), e.mean()))
