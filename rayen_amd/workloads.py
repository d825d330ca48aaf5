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


# ---------------------------------------------------------------------------------------------------------------
# Config 5 with the STRUCTURE of the reference's generator (examples/scripts/matlab/traj_planning_in_corridor.m:56-104,
# getCorridorAndParamsSpline.m:23-48, getABVerticesgivenP1P2.m:9-80, MyClampedUniformSpline.m:26-37, 82-99, 682-736).
# The data file itself (corridor_dim3.mat) is an absent git-LFS pointer and the MATLAB toolchain (casadi, MINVO basis
# matrices, vert2lcon) is not vendored, so this is a restatement, not a reproduction:
#   * 7 way-points -> 6 regions; a region = convex hull of 16 points drawn within radius 1 of the 8 vertices of a box of
#     half-side 1 around its segment, outside the box (22-26 faces each; scipy's hull instead of vert2lcon, numpy's
#     generator instead of MATLAB's rng(2));
#   * trajectory = clamped uniform cubic B-spline, 12 intervals on [0, 15] -> 15 control points in R^3: k = 45;
#   * corridor rows: the four control points of interval j lie in region ceil(j / 2) -- in the BEZIER basis (control
#     points of the interval by blossoming), where the reference uses the MINVO basis, whose matrices are not in the
#     tree; both enclose the interval, so the rows have the same form A_r (M_j Q) <= b_r;
#   * 15 equality rows: position at t0, velocity and acceleration zero at t0 and tf -> n = 30;
#   * 72 quadratics ||c||^2 <= ||limit||^2 on the (3 + 2 + 1) velocity / acceleration / jerk control points of every
#     interval (B-spline basis of the derivative splines), limits 4 / 6 / 50 per axis: P = 2 C'C of rank 3;
#   * exact duplicates among the rows (the last Bezier point of an interval is the first of the next) are dropped, the
#     LP-based redundancy removal of rayen/constraints.py:256-286 is NOT applied (so that the golden vectors can come
#     from the reference itself, which cannot run its LPs here): 1.1 k rows, as SURVEY.md 8(d) estimated;
#   * the interior point is computed here once (max-margin conic program of rayen_amd/conic.py) and handed over as y0.
# ---------------------------------------------------------------------------------------------------------------
def _corridor_regions(rng):
    from scipy.spatial import ConvexHull
    way = 3.0 * np.array([[0, 1, 2, 3, 4, 3, 0], [0, 1, 1, 2, 4, 4, 4], [0, 1, 1, 1, 4, 1, 0]], dtype=float)
    regions = []
    for i in range(way.shape[1] - 1):
        p1, p2 = way[:, i], way[:, i + 1]
        h = np.linalg.norm(p2 - p1)
        zb = (p2 - p1) / h
        xb = np.cross(np.array([0.0, 1.0, 0.0]), zb)
        xb /= np.linalg.norm(xb)
        frame = np.stack([xb, np.cross(zb, xb), zb], axis=1)
        verts = [p1 + frame @ np.array([sx, sy, sz]) for sz in (0.0, h) for sx in (1.0, -1.0) for sy in (1.0, -1.0)]
        a_box = np.concatenate([frame.T, -frame.T])
        b_box = np.array([max(a_box[r] @ vtx for vtx in verts) for r in range(6)])
        pts = []
        for vtx in verts:
            kept = 0
            while kept < 2:
                d = rng.normal(size=3)
                d *= rng.uniform() ** (1.0 / 3.0) / np.linalg.norm(d)       # uniform in the unit ball
                if np.any(a_box @ (vtx + d) - b_box > 0.0):                 # outside the box
                    pts.append(vtx + d)
                    kept += 1
        hull = ConvexHull(np.array(pts))
        regions.append((hull.equations[:, :3].copy(), -hull.equations[:, 3].copy(), np.array(pts)))
    return regions


def _bezier_of_interval(knots, p, i):
    """``[p+1, p+1]`` matrix taking the control points ``Q_{i-p..i}`` of a degree-``p`` B-spline to the Bezier
    points of its interval ``[knots[i], knots[i+1]]``: point ``m`` is the blossom at ``knots[i]`` (p - m times),
    ``knots[i+1]`` (m times), evaluated by the de Boor recursion."""
    out = np.zeros((p + 1, p + 1))
    for m in range(p + 1):
        args = [knots[i]] * (p - m) + [knots[i + 1]] * m
        d = np.eye(p + 1)                                   # row j: coefficients of Q_{i-p+j}
        for r in range(1, p + 1):
            nxt = d.copy()
            for j in range(i - p + r, i + 1):
                a = (args[r - 1] - knots[j]) / (knots[j + p - r + 1] - knots[j])
                nxt[j - (i - p)] = (1.0 - a) * d[j - 1 - (i - p)] + a * d[j - (i - p)]
            d = nxt
        out[m] = d[p]
    return out


_CORRIDOR_CACHE = {}


def corridor_spline(seed=0):
    """Config 5: the corridor trajectory set restated from the reference's MATLAB generator (see the block comment above):
    k = 45, 15 equalities (n = 30), ~1.1 k corridor rows, 72 rank-3 quadratics, interior point computed once."""
    if seed in _CORRIDOR_CACHE:
        return {key: (list(val) if isinstance(val, list) else (None if val is None else np.array(val)))
                if key != "do_preprocessing_linear" else val for key, val in _CORRIDOR_CACHE[seed].items()}
    rng = np.random.default_rng(1000 + seed)
    regions = _corridor_regions(rng)
    p, n_seg, dim, t0, tf = 3, 12, 3, 0.0, 15.0
    n_cp = n_seg + p                                                      # 15
    k = n_cp * dim
    dt = (tf - t0) / n_seg
    knots = np.concatenate([np.full(p + 1, t0), t0 + dt * np.arange(1, n_seg), np.full(p + 1, tf)])

    def sel(l):                                                           # 3 x k: picks control point l
        out = np.zeros((dim, k))
        out[:, dim * l: dim * l + dim] = np.eye(dim)
        return out

    vel = [p * (sel(l + 1) - sel(l)) / (knots[l + p + 1] - knots[l + 1]) for l in range(n_cp - 1)]
    acc = [(p - 1) * (vel[l + 1] - vel[l]) / (knots[l + p + 1] - knots[l + 2]) for l in range(n_cp - 2)]
    jerk = [(p - 2) * (acc[l + 1] - acc[l]) / (knots[l + p + 1] - knots[l + 3]) for l in range(n_cp - 3)]

    raw = _empty(k)
    rows_a, rows_b = [], []
    for j in range(1, n_seg + 1):
        i = p + j - 1                                                     # interval [knots[i], knots[i + 1]]
        bez = _bezier_of_interval(knots, p, i)
        a_r, b_r, _ = regions[(j + 1) // 2 - 1]
        for m in range(p + 1):
            point = sum(bez[m, c] * sel(i - p + c) for c in range(p + 1))  # 3 x k
            rows_a.append(a_r @ point)
            rows_b.append(b_r.reshape(-1, 1))
    a1, b1 = np.concatenate(rows_a), np.concatenate(rows_b)
    _, keep = np.unique(np.round(np.concatenate([a1, b1], axis=1), 12), axis=0, return_index=True)
    keep = np.sort(keep)
    raw["A1"], raw["b1"] = a1[keep], b1[keep]
    start = regions[0][2].mean(axis=0).reshape(dim, 1)
    raw["A2"] = np.concatenate([sel(0), vel[0], vel[-1], acc[0], acc[-1]])
    raw["b2"] = np.concatenate([start, np.zeros((4 * dim, 1))])
    for j in range(1, n_seg + 1):
        for c_mat, lim in ([(vel[j - 1 + c], 4.0) for c in range(3)] + [(acc[j - 1 + c], 6.0) for c in range(2)]
                           + [(jerk[j - 1], 50.0)]):
            raw["P"].append(2.0 * c_mat.T @ c_mat)
            raw["q"].append(np.zeros((k, 1)))
            raw["r"].append(np.array([[-dim * lim * lim]]))
    # interior point: the package's own solver-free search, once
    lc = constraints.LinearConstraint(raw["A1"], raw["b1"], raw["A2"], raw["b2"])
    qcs = [constraints.ConvexQuadraticConstraint(P, q, r, do_checks_P=False) for P, q, r in zip(raw["P"], raw["q"], raw["r"])]
    found = constraints.ConvexConstraints(lc=lc, qcs=qcs, y0=None, do_preprocessing_linear=False)
    raw["y0"] = np.array(found.y0).reshape(k, 1)
    _CORRIDOR_CACHE[seed] = raw
    return corridor_spline(seed)


CONFIGS = {
    # name: (builder, kwargs, batch named in BASELINE.json, input range)
    "c1": (cube, {}, 500, 5.0),
    "c2": (random_lin_quad_soc, dict(k=16, m=32, n_quad=2, n_soc=0), 4096, 1.0),
    "c3": (random_lin_quad_soc, dict(k=64, m=128, n_quad=4, n_soc=2), 262144, 1.0),
    "c4": (random_lmi, dict(k=10, r=20), 16384, 1.0),
    "c5": (corridor_spline, {}, 2097152, 1.0),
    "c5r": (corridor_like, {}, 2097152, 1.0),     # the random stand-in of rounds 1-2 (288 rows), kept for comparison
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


def violation_report(cs, y):
    """Where the largest residual of a batch ``y [B,k]`` sits and how it compares with what ROUNDING ``y`` TO FP32 alone
    can cause: per family the worst residual (``ConvexConstraints.getResiduals``) and the worst ratio
    residual / (2^-24 x the sum of the absolute values of the terms the residual is made of) -- a feasible point of a
    row ``a'y <= b`` rounded to fp32 can leave up to 2^-24 (|a|'|y|) of residual, a quadratic
    2^-24 (|y|'|P||y| + |q|'|y|), and so on.  A ratio of a few units says the output is feasible to the working precision of
    its storage format; the absolute number says how large the set's coefficients are (host code, fp64, no oracle)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1, cs.k)
    ay = np.abs(y)
    u = 2.0 ** -24
    fam = {}
    if cs.has_linear_ineq_constraints:
        r = y @ cs.lc.A1.T - cs.lc.b1.T
        t = ay @ np.abs(cs.lc.A1).T + np.abs(cs.lc.b1).T
        fam["lin_ineq"] = (r, t)
    if cs.has_linear_eq_constraints:
        r = np.abs(y @ cs.lc.A2.T - cs.lc.b2.T)
        t = ay @ np.abs(cs.lc.A2).T + np.abs(cs.lc.b2).T
        fam["lin_eq"] = (r, t)
    if cs.has_quadratic_constraints:
        r = np.stack([0.5 * np.einsum("bi,ij,bj->b", y, qc.P, y) + y @ qc.q[:, 0] + qc.r[0, 0] for qc in cs.qcs], axis=1)
        t = np.stack([np.einsum("bi,ij,bj->b", ay, np.abs(qc.P), ay) + ay @ np.abs(qc.q[:, 0]) for qc in cs.qcs], axis=1)
        fam["quad"] = (r, t)
    if cs.has_soc_constraints:
        r = np.stack([np.linalg.norm(y @ soc.M.T + soc.s.T, axis=1) - (y @ soc.c[:, 0] + soc.d[0, 0]) for soc in cs.socs], axis=1)
        t = np.stack([np.linalg.norm(ay @ np.abs(soc.M).T, axis=1) + ay @ np.abs(soc.c[:, 0]) for soc in cs.socs], axis=1)
        fam["soc"] = (r, t)
    if cs.has_lmi_constraints:
        F = np.stack(cs.lmic.all_F[:-1], axis=0)
        H = np.einsum("ba,ajk->bjk", y, F) + cs.lmic.all_F[-1][None]
        r = -np.linalg.eigvalsh(H)[:, :1]
        t = np.einsum("ba,a->b", ay, np.linalg.norm(F, ord=2, axis=(1, 2)))[:, None]
        fam["lmi"] = (r, t)
    out = {}
    for name, (r, t) in fam.items():
        out[name] = {"max_residual": float(np.max(r)),
                     "over_f32_rounding_of_y": float(np.max(np.maximum(r, 0.0) / np.maximum(u * t, 1e-300)))}
    worst = max(out, key=lambda f: out[f]["max_residual"])
    return {"family_of_max": worst, "per_family": out}
