"""Generate the golden vectors under ``tests/golden/`` from the REAL reference.

Runs only in the build container, where the reference is mounted read-only at
``/root/reference``; never on the GPU box (nothing in ``tests/`` imports this
file).  Only its *outputs* (``*.npz``: constraint data, inputs, and the
reference's own outputs) are committed.

The reference's RAYEN path imports ``cvxpy``, ``cvxpylayers``, ``cdd`` and
``colorama`` at module scope but, when an interior point ``y0`` is supplied and
linear preprocessing is off, only *constructs* (never solves) cvxpy problems
(SURVEY.md §8c).  Those four packages are absent here, so this script registers
inert stand-in modules before importing the reference; the numerical path that
is exercised (numpy/scipy/torch) is the reference's own, unmodified.

    python tests/golden/make_golden.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# --------------------------------------------------------------------------- inert stand-ins
class _Inert:
    """Absorbs every operator / call / attribute; used for cvxpy expressions."""
    __array_ufunc__ = None  # make ``ndarray @ inert`` defer to us

    def __init__(self, *a, **k):
        pass

    def _same(self, *a, **k):
        return _Inert()

    __call__ = __getitem__ = _same
    for _op in ("add", "radd", "sub", "rsub", "mul", "rmul", "matmul", "rmatmul", "truediv",
                "rtruediv", "neg", "pos", "le", "ge", "eq", "lt", "gt", "rshift", "rrshift",
                "lshift", "rlshift", "pow"):
        locals()[f"__{_op}__"] = _same
    del _op
    __hash__ = object.__hash__

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


def _install_stubs():
    cp = types.ModuleType("cvxpy")
    for name in ("Variable", "Parameter", "Minimize", "Maximize", "Problem", "sum_squares",
                 "quad_form", "norm"):
        setattr(cp, name, _Inert)
    cp.installed_solvers = lambda: ["SCS"]
    sys.modules["cvxpy"] = cp

    cvxl = types.ModuleType("cvxpylayers")
    cvxl_t = types.ModuleType("cvxpylayers.torch")
    cvxl_t.CvxpyLayer = _Inert
    cvxl.torch = cvxl_t
    sys.modules["cvxpylayers"] = cvxl
    sys.modules["cvxpylayers.torch"] = cvxl_t

    sys.modules["cdd"] = types.ModuleType("cdd")

    col = types.ModuleType("colorama")

    class _Blank:
        def __getattr__(self, name):
            return ""
    col.Fore = col.Back = col.Style = _Blank()
    col.init = lambda *a, **k: None
    sys.modules["colorama"] = col


_install_stubs()
sys.path.insert(0, REF)                                   # the reference's own `rayen` package wins
sys.path.insert(0, os.path.join(REF, "examples"))
sys.path.append(REPO)                                      # rayen_amd.workloads (raw synthetic data only)

from rayen import constraints as ref_constraints          # noqa: E402  (the reference)
from rayen import constraint_module as ref_module         # noqa: E402
import examples_sets as ref_examples                      # noqa: E402
from rayen_amd import workloads                           # noqa: E402  (raw synthetic data only)


# interior points that SURVEY.md §8(c) verified for the 15 sets of examples_sets.py:94-194
EXAMPLE_Y0 = {
    0: [1 / 3, 1 / 3, 1 / 3], 1: [1 / 3, 1 / 3, 1 / 3], 7: [1 / 3, 1 / 3, 1 / 3],
    2: [0, 0, 0], 3: [0, 0, 1], 10: [0, 0, 1], 11: [0, 0, 1],
    4: [0.5, 0.5], 5: [0.5, 0.5], 6: [0.45, 0.275, 0.275], 8: [5, 3],
    9: [0, 0, 1 / 3], 12: [1, 0, 1], 13: [0.5, 0, 0.8], 14: [1.1, 0.1, 0.1],
}


def _raw_from_ref_cs(cs):
    """User-level data of a reference ConvexConstraints object, as plain arrays."""
    raw = dict(A1=None, b1=None, A2=None, b2=None)
    if cs.lc is not None:
        raw.update(A1=cs.lc.A1, b1=cs.lc.b1, A2=cs.lc.A2, b2=cs.lc.b2)
    raw["P"] = [qc.P for qc in cs.qcs]
    raw["q"] = [qc.q for qc in cs.qcs]
    raw["r"] = [qc.r for qc in cs.qcs]
    raw["M"] = [s.M for s in cs.socs]
    raw["s"] = [s.s for s in cs.socs]
    raw["c"] = [s.c for s in cs.socs]
    raw["d"] = [s.d for s in cs.socs]
    raw["F"] = list(cs.lmic.all_F) if cs.lmic is not None else []
    return raw


def _ref_cs_from_raw(raw):
    lc = None
    if raw["A1"] is not None or raw["A2"] is not None:
        lc = ref_constraints.LinearConstraint(raw["A1"], raw["b1"], raw["A2"], raw["b2"])
    qcs = [ref_constraints.ConvexQuadraticConstraint(P, q, r, do_checks_P=False)
           for P, q, r in zip(raw["P"], raw["q"], raw["r"])]
    socs = [ref_constraints.SOCConstraint(M, s, c, d)
            for M, s, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"])]
    lmic = ref_constraints.LMIConstraint([np.array(F) for F in raw["F"]]) if len(raw["F"]) else None
    return ref_constraints.ConvexConstraints(lc=lc, qcs=qcs, socs=socs, lmic=lmic, y0=raw["y0"],
                                             do_preprocessing_linear=False)


def _example_cs(index):
    """examples_sets.getExample(index) with the interior point injected (no solver runs)."""
    y0 = np.array(EXAMPLE_Y0[index], dtype=np.float64).reshape(-1, 1)
    real_init = ref_constraints.ConvexConstraints.__init__

    def patched(self, *a, **kw):
        kw["y0"] = y0
        kw["do_preprocessing_linear"] = False
        real_init(self, *a, **kw)

    ref_constraints.ConvexConstraints.__init__ = patched
    try:
        return ref_examples.getExample(index)
    finally:
        ref_constraints.ConvexConstraints.__init__ = real_init


def _run_reference(cs, x32):
    """Reference forward + kappa at fp32 and fp64 on the same (fp32-representable) inputs."""
    out = {}
    for tag, dtype in (("32", torch.float32), ("64", torch.float64)):
        torch.set_default_dtype(dtype)
        try:
            layer = ref_module.ConstraintModule(cs, method="RAYEN", create_map=False)
            layer.eval()
            x = torch.tensor(x32).to(dtype)
            with torch.no_grad():
                y = layer(x)
                v_bar = torch.nn.functional.normalize(x[:, 0:cs.n, 0:1], dim=1)
                kappa_bar = layer.computeKappa(v_bar)
            out["y" + tag] = y.numpy()[:, :, 0]
            out["kappa_bar" + tag] = kappa_bar.numpy()[:, 0, 0]
            for name in ("D", "all_phi", "all_delta", "L"):
                if hasattr(layer, name):
                    out[f"buf_{name}{tag}"] = getattr(layer, name).numpy()
            # the RAYEN_old head on the same directions plus a step column beta (CM:460-466)
            old = ref_module.ConstraintModule(cs, method="RAYEN_old", create_map=False)
            old.eval()
            gen = torch.Generator().manual_seed(77)
            beta = torch.empty(x32.shape[0], 1, 1, dtype=torch.float32).uniform_(-2.0, 2.0, generator=gen)
            with torch.no_grad():
                y_old = old(torch.cat((x, beta.to(dtype)), dim=1))
            out["beta"] = beta.numpy()
            out["y_old" + tag] = y_old.numpy()[:, :, 0]
        finally:
            torch.set_default_dtype(torch.float32)
    return out


def _save(name, raw, cs, x32, ref_out):
    data = {"x": x32}
    for key in ("A1", "b1", "A2", "b2"):
        if raw[key] is not None:
            data["raw_" + key] = np.asarray(raw[key], dtype=np.float64)
    for key in ("P", "q", "r", "M", "s", "c", "d", "F"):
        if len(raw[key]):
            data["raw_" + key] = np.stack([np.asarray(a, dtype=np.float64) for a in raw[key]])
    data["raw_y0"] = np.asarray(cs.y0, dtype=np.float64)
    for key in ("A_p", "b_p", "NA_E", "yp", "z0"):
        data["cs_" + key] = np.asarray(getattr(cs, key), dtype=np.float64)
    data.update(ref_out)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **data)
    print(f"{name}: k={cs.k} n={cs.n} B={x32.shape[0]}  -> {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    torch.manual_seed(0)
    # ---- the 15 example sets, 500 samples U(-5,5) (examples/test_layer.py:74-75) + two interior rows
    for index in range(15):
        if sys.argv[1:] and f"example_{index:02d}" not in sys.argv[1:]:
            continue
        cs = _example_cs(index)
        gen = torch.Generator().manual_seed(100 + index)
        x = torch.empty(500, cs.n, 1, dtype=torch.float32).uniform_(-5.0, 5.0, generator=gen)
        x = torch.cat((x, torch.zeros(1, cs.n, 1), torch.full((1, cs.n, 1), 1e-3 / np.sqrt(cs.n))))
        raw = _raw_from_ref_cs(cs)
        raw["y0"] = cs.y0
        _save(f"example_{index:02d}", raw, cs, x.numpy(), _run_reference(cs, x.numpy()))

    # ---- B=256 slices of configs 2-5 (time_analysis.py generators, y0 as in rayen_amd.workloads)
    slices = {
        "config_c2": workloads.make_raw("c2", seed=2),
        "config_c3": workloads.make_raw("c3", seed=3),
        "config_c4": workloads.make_raw("c4", seed=4),
        "config_c5": workloads.corridor_like(k=45, n_eq=15, m=64, n_quad=8, rank=3, seed=5),
        # small mixed set that exercises every family at once with equalities
        "config_mixed": None,
    }
    rng = np.random.default_rng(7)
    mixed = workloads.random_lin_quad_soc(k=12, m=20, n_quad=2, n_soc=2, r_M=5, seed=7)
    lmi = workloads.random_lmi(k=12, r=6, seed=8)
    mixed["F"] = lmi["F"]
    mixed["A2"] = rng.uniform(-1, 1, size=(3, 12))
    mixed["b2"] = np.zeros((3, 1))
    slices["config_mixed"] = mixed
    # config 5 with the structure of the reference's corridor generator (rayen_amd.workloads.corridor_spline: 1050 rows,
    # 15 equalities, 72 rank-3 quadratics; the interior point found there is handed to the reference as y0).  Added in
    # round 3 behind the older slices: their seeds -- and files -- are unchanged.
    slices["config_c5s"] = workloads.make_raw("c5", seed=0)

    only = set(sys.argv[1:])
    for seed, (name, raw) in enumerate(slices.items()):
        if only and name not in only:
            continue
        cs = _ref_cs_from_raw(raw)
        gen = torch.Generator().manual_seed(500 + seed)
        x = torch.empty(256, cs.n, 1, dtype=torch.float32).uniform_(-1.0, 1.0, generator=gen)
        _save(name, raw, cs, x.numpy(), _run_reference(cs, x.numpy()))


if __name__ == "__main__":
    main()
