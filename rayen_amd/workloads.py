"""Synthetic constraint sets for the BASELINE.json configurations.

The generators follow the recipes of the reference's timing sweep
(``examples/scripts/time_analysis.py``): random ``A1 ~ U(-1,1)``, ``b1 ~ U(0.1,1)``
(time_analysis.py:68-72); ``P = T T'``, ``q ~ U(-1,1)``, ``r ~ U(-1,0)``
(time_analysis.py:91-99); ``M,s,c ~ U(-1,1)``, ``d = ||s|| + 0.5``
(time_analysis.py:127-132); symmetric ``F_i`` and ``F_k = T T' + 0.5 I``
(time_analysis.py:166-175); interior point ``y0 = 0``.  Config 1 is the unit cube
of ``examples/examples_sets.py:14-29`` with ``y0 = (.5,.5,.5)``.  Config 5 stands in
for the absent ``corridor_dim3.mat`` (git-LFS pointer) with the structure
SURVEY.md §8(d) recovers from the MATLAB generator: k=45, 15 equalities (n=30),
288 inequality rows, 72 rank-3 quadratics ``P = 2 C'C``.

All generators return a *raw* dict of fp64 numpy arrays (keys ``A1,b1,A2,b2``,
lists ``P,q,r``, ``M,s,c,d``, ``F`` and ``y0``, ``do_preprocessing_linear``) so that the
same data can be fed to this package's classes (``build_constraints``) and, in
the build container, to the reference (``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import numpy as np

from . import constraints


def _empty(k):
    return dict(A1=None, b1=None, A2=None, b2=None, P=[], q=[], r=[], M=[], s=[], c=[], d=[],
                F=[], y0=np.zeros((k, 1)), do_preprocessing_linear=False)


def cube(seed=0):
    """Config 1: ``0 <= y <= 1`` in R^3, interior point the centre."""
    raw = _empty(3)
    raw["A1"] = np.concatenate((np.eye(3), -np.eye(3)), axis=0)
    raw["b1"] = np.array([[1.0], [1.0], [1.0], [0.0], [0.0], [0.0]])
    raw["y0"] = np.full((3, 1), 0.5)
    return raw


def random_lin_quad_soc(k, m, n_quad, n_soc, r_M=None, seed=0):
    """Configs 2/3: ``m`` linear rows, ``n_quad`` dense quadratics, ``n_soc`` SOCs, ``y0 = 0``."""
    rng = np.random.default_rng(seed)
    r_M = k if r_M is None else r_M
    raw = _empty(k)
    if m > 0:
        raw["A1"] = rng.uniform(-1.0, 1.0, size=(m, k))
        raw["b1"] = rng.uniform(0.1, 1.0, size=(m, 1))
    for _ in range(n_quad):
        T = rng.uniform(-1.0, 1.0, size=(k, k))
        raw["P"].append(T @ T.T)
        raw["q"].append(rng.uniform(-1.0, 1.0, size=(k, 1)))
        raw["r"].append(rng.uniform(-1.0, 0.0, size=(1, 1)))
    for _ in range(n_soc):
        s = rng.uniform(-1.0, 1.0, size=(r_M, 1))
        raw["M"].append(rng.uniform(-1.0, 1.0, size=(r_M, k)))
        raw["s"].append(s)
        raw["c"].append(rng.uniform(-1.0, 1.0, size=(k, 1)))
        raw["d"].append(np.linalg.norm(s) + np.array([[0.5]]))
    return raw


def random_lmi(k, r, seed=0):
    """Config 4: one LMI with ``k`` symmetric ``r x r`` generators and an SPD offset."""
    rng = np.random.default_rng(seed)
    raw = _empty(k)
    for _ in range(k):
        T = rng.uniform(-1.0, 1.0, size=(r, r))
        raw["F"].append((T + T.T) / 2)
    T = rng.uniform(-1.0, 1.0, size=(r, r))
    raw["F"].append(T @ T.T + 0.5 * np.eye(r))
    return raw


def corridor_like(k=45, n_eq=15, m=288, n_quad=72, rank=3, seed=0):
    """Config 5 stand-in: equalities + inequalities + low-rank quadratics around a random ``y0``."""
    rng = np.random.default_rng(seed)
    raw = _empty(k)
    y0 = rng.uniform(-1.0, 1.0, size=(k, 1))
    raw["y0"] = y0
    raw["A2"] = rng.uniform(-1.0, 1.0, size=(n_eq, k))
    raw["b2"] = raw["A2"] @ y0
    raw["A1"] = rng.uniform(-1.0, 1.0, size=(m, k))
    raw["b1"] = raw["A1"] @ y0 + rng.uniform(0.1, 1.0, size=(m, 1))
    for _ in range(n_quad):
        C = rng.uniform(-1.0, 1.0, size=(rank, k))
        P = 2.0 * C.T @ C
        q = rng.uniform(-1.0, 1.0, size=(k, 1))
        g0 = 0.5 * y0.T @ P @ y0 + q.T @ y0
        raw["P"].append(P)
        raw["q"].append(q)
        raw["r"].append(-g0 - rng.uniform(0.1, 1.0, size=(1, 1)))
    return raw


CONFIGS = {
    # name: (builder, kwargs, batch named in BASELINE.json, input range)
    "c1": (cube, {}, 500, 5.0),
    "c2": (random_lin_quad_soc, dict(k=16, m=32, n_quad=2, n_soc=0), 4096, 1.0),
    "c3": (random_lin_quad_soc, dict(k=64, m=128, n_quad=4, n_soc=2), 262144, 1.0),
    "c4": (random_lmi, dict(k=10, r=20), 16384, 1.0),
    "c5": (corridor_like, {}, 2097152, 1.0),
}


def make_raw(name, seed=0):
    builder, kwargs, _, _ = CONFIGS[name]
    return builder(seed=seed, **kwargs)


def build_constraints(raw):
    """Raw dict -> :class:`rayen_amd.constraints.ConvexConstraints` with the explicit ``y0``."""
    lc = None
    if raw["A1"] is not None or raw["A2"] is not None:
        lc = constraints.LinearConstraint(raw["A1"], raw["b1"], raw["A2"], raw["b2"])
    qcs = [constraints.ConvexQuadraticConstraint(P, q, r, do_checks_P=False)
           for P, q, r in zip(raw["P"], raw["q"], raw["r"])]
    socs = [constraints.SOCConstraint(M, s, c, d)
            for M, s, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"])]
    lmic = constraints.LMIConstraint(list(raw["F"])) if len(raw["F"]) else None
    return constraints.ConvexConstraints(lc=lc, qcs=qcs, socs=socs, lmic=lmic, y0=raw["y0"],
                                         do_preprocessing_linear=raw["do_preprocessing_linear"])


def algorithmic_work(cs):
    """(bytes, flops) per projection in fp32, SURVEY.md §8(d) formulas.

    Two refinements keep the figure a lower bound of the arithmetic any exact evaluation needs (so that
    a roofline fraction can never exceed 1 by construction): a quadratic of rank ``rho < n/2`` is priced
    in its factored form ``||U v||`` (``2 rho n`` instead of ``2 n^2``; config 5's ``P = 2 C'C`` has rank 3),
    and products are priced in the subspace dimension ``n`` the kernels work in (``n = k`` unless the set
    has equalities).
    """
    import numpy as np
    k, n = cs.k, cs.n
    m = cs.A_p.shape[0]
    flops = 2 * m * n + 2 * k * n
    for qc in cs.qcs:
        rank = int(np.linalg.matrix_rank(qc.P))
        flops += min(2 * n * n, 2 * rank * n + 2 * rank) + 4 * n
    for soc in cs.socs:
        r_M = soc.M.shape[0]
        flops += 2 * r_M * n + 2 * n + 4 * r_M
    if cs.has_lmi_constraints:
        r = cs.lmic.all_F[0].shape[0]
        flops += 2 * n * r * r + (4 * r ** 3) // 3
    return 4 * (n + k), flops
