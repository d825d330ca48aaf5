"""Shared test helpers: golden-case loading and conversions between the three views of a case.

* ``raw``   -- user-level constraint data (A1,b1,A2,b2, lists P,q,r / M,s,c,d / F, y0)
* ``csd``   -- the preprocessed fields the oracle's ``precompute`` reads
* ``cs``    -- this package's ``ConvexConstraints`` built from ``raw``
"""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, "*.npz")))


def load_golden(name):
    """(raw, csd, z).  ``z`` is a dict of the fixture's arrays.  Where the reference's OWN fp32 forward is not finite
    (config_c5s: its ``sqrt(rho' delta rho)`` takes a slightly negative radicand on 119 of 256 directions of the
    corridor set -- the hazard examples/main.py:288 trains in fp64 to avoid; generated under ``python -O``, without
    which CM:342-381's asserts stop the reference first) the fp32 entries ``y32 / kappa_bar32 / y_old32`` of those rows
    hold the reference's fp64 output rounded to fp32 -- the truth a finite fp32 implementation is held to -- and
    ``z["nan_rows32"]`` marks them (all False for every other fixture)."""
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    nan_rows = ~np.isfinite(z["y32"]).all(axis=1)
    z["nan_rows32"] = nan_rows
    if nan_rows.any():
        for key in ("y", "kappa_bar", "y_old"):
            if key + "32" in z:
                fixed = z[key + "32"].copy()
                fixed[nan_rows] = z[key + "64"][nan_rows].astype(np.float32)
                z[key + "32"] = fixed
    raw = dict(A1=None, b1=None, A2=None, b2=None, do_preprocessing_linear=False)
    for key in ("A1", "b1", "A2", "b2"):
        if "raw_" + key in z:
            raw[key] = z["raw_" + key]
    for key in ("P", "q", "r", "M", "s", "c", "d", "F"):
        raw[key] = list(z["raw_" + key]) if "raw_" + key in z else []
    raw["y0"] = z["raw_y0"]
    csd = {key: z["cs_" + key] for key in ("A_p", "b_p", "NA_E", "yp", "z0")}
    csd["y0"] = raw["y0"]
    for key in ("P", "q", "r", "M", "s", "c", "d", "F"):
        csd[key] = raw[key]
    return raw, csd, z


def csd_from_cs(cs):
    """Oracle input dict from a (this package's) ConvexConstraints object."""
    csd = {key: getattr(cs, key) for key in ("A_p", "b_p", "NA_E", "yp", "z0", "y0")}
    csd["P"] = [qc.P for qc in cs.qcs]
    csd["q"] = [qc.q for qc in cs.qcs]
    csd["r"] = [qc.r for qc in cs.qcs]
    csd["M"] = [s.M for s in cs.socs]
    csd["s"] = [s.s for s in cs.socs]
    csd["c"] = [s.c for s in cs.socs]
    csd["d"] = [s.d for s in cs.socs]
    csd["F"] = list(cs.lmic.all_F) if cs.lmic is not None else []
    return csd


def raw_from_cs(cs):
    raw = dict(A1=None, b1=None, A2=None, b2=None)
    if cs.lc is not None:
        raw.update(A1=cs.lc.A1, b1=cs.lc.b1, A2=cs.lc.A2, b2=cs.lc.b2)
    raw.update({k: v for k, v in csd_from_cs(cs).items() if k in "PqrMscdF"})
    return raw


def rel_err_rows(y, y_ref):
    """Per-sample inf-norm relative error (the 1e-5 parity metric of BASELINE.json)."""
    y = np.asarray(y, dtype=np.float64)
    y_ref = np.asarray(y_ref, dtype=np.float64)
    num = np.max(np.abs(y - y_ref), axis=1)
    den = np.maximum(np.max(np.abs(y_ref), axis=1), 1e-30)
    return num / den


def kink_mask(oracle, cs, x, rel_gap, method="RAYEN"):
    """Samples at which the layer's gradient is legitimately discontinuous, identified on the fp64 oracle:

    * two candidates of the max in ``computeKappa`` tie within ``rel_gap`` of kappa (the arg-max may flip; this
      includes the two largest LMI eigenvalues, whose eigenvector derivative diverges as they meet),
    * ``kappa(v)`` within ``rel_gap`` of 1 (the RAYEN head switches between clipped and unclipped), or of 0
      (relu corner).

    ``x [B, >=n, 1]`` as fed to the layer.  Returns a boolean numpy array ``[B]``."""
    import torch
    buf = oracle.precompute(csd_from_cs(cs), torch.float64)
    n = cs.n
    v = x[:, 0:n, 0:1].double().cpu()
    norm = torch.linalg.vector_norm(v, dim=(1, 2)).clamp_min(1e-300)
    cand = oracle.compute_kappa(buf, v / norm.reshape(-1, 1, 1), terms=True) * norm.reshape(-1, 1)
    cand = torch.nan_to_num(cand, nan=0.0)
    top2 = torch.topk(torch.cat((cand, torch.zeros(cand.shape[0], 2, dtype=cand.dtype)), dim=1), 2, dim=1).values
    kappa = top2[:, 0].clamp_min(0.0)
    scale = kappa.clamp_min(1e-300)
    tie = (top2[:, 0] - top2[:, 1]) <= rel_gap * scale
    tie &= kappa > 0
    # relu corner: the largest candidate sits next to zero.  A candidate that IS zero has no gradient and is no
    # corner: the placeholder row 0z <= 1 of a set without linear constraints (rayen/constraints.py:386-388), or a
    # quadratic / cone root the reference has already clamped with relu (an unbounded direction, CM:348)
    cmax = cand.max(dim=1).values
    near_zero = (cmax.abs() <= rel_gap * norm) & (cmax != 0) & (norm > 0)
    kink = tie | near_zero
    if buf["all_F"].ndim == 3:
        # eigenvector sensitivity: an error dS in the pencil matrix turns the top eigenvector by ~|dS| / (lam1 - lam2)
        # and |dS| scales with the SPECTRAL RADIUS, not with kappa = lam1 -- a gap that is small against the radius
        # is a kink for the gradient even when it is not small against kappa
        rho_v = buf["NA_E"] @ (v / norm.reshape(-1, 1, 1))
        S = torch.einsum("ajk,ial->ijk", [buf["all_F"][0:-1], rho_v])
        lam = torch.linalg.eigvalsh(buf["L"].T @ (-S) @ buf["L"]) * norm.reshape(-1, 1)
        if lam.shape[1] >= 2:
            lmi_on_top = (lam[:, -1] >= cand.max(dim=1).values * (1.0 - rel_gap)) & (kappa > 0)
            kink |= lmi_on_top & ((lam[:, -1] - lam[:, -2]) <= 10.0 * rel_gap * lam.abs().amax(dim=1))
    if method == "RAYEN":
        kink |= (kappa - 1.0).abs() <= rel_gap
    kink = kink.numpy()
    # square-root singularities of the ACTIVE constraint: a ray tangent to a cone (the quadratic a'x^2 + b'x + c' has
    # a double root: d kappa / d v = -(...) / (2 a' kappa + b') diverges, rayen/constraint_module.py:339-348, 392-396)
    # or a quadratic whose radicand rho' delta rho vanishes (CM:374).  Identified by the size of the vanishing quantity
    # relative to its terms, below sqrt(rel_gap) (the gradient error grows like rounding / that ratio).
    m = buf["D"].shape[0]
    n_quad = buf["all_P"].shape[0] if buf["all_P"].ndim == 3 else 0
    n_soc = buf["all_M"].shape[0] if buf["all_M"].ndim == 3 else 0
    arg = torch.argmax(cand, dim=1).numpy()
    rho = (buf["NA_E"] @ (v / norm.reshape(-1, 1, 1)))[:, :, 0].numpy()          # unit directions in y space
    kap_bar = (kappa / norm).numpy()
    thr = float(np.sqrt(rel_gap))
    for j in range(n_soc):
        idx = np.flatnonzero(arg == m + n_quad + j)
        if idx.size == 0:
            continue
        M, s_, c, d = (buf[key][j].numpy() for key in ("all_M", "all_s", "all_c", "all_d"))
        y0 = buf["y0"].numpy()
        beta, tau = (M @ y0 + s_)[:, 0], float((c.T @ y0 + d).item())
        r = rho[idx]
        cr = r @ c[:, 0]
        b_p = 2.0 * (r @ M.T) @ beta - 2.0 * cr * tau
        a_p = float(beta @ beta - tau * tau)
        den = 2.0 * a_p * kap_bar[idx] + b_p
        size = np.abs(2.0 * a_p * kap_bar[idx]) + np.abs(b_p) + 1e-300
        kink[idx[np.abs(den) <= thr * size]] = True
    for i in range(n_quad):
        idx = np.flatnonzero(arg == m + i)
        if idx.size == 0:
            continue
        delta = buf["all_delta"][i].numpy()
        r = rho[idx]
        rad = np.einsum("bi,ij,bj->b", r, delta, r)
        size = np.linalg.norm(delta, 2) * np.einsum("bi,bi->b", r, r) + 1e-300
        kink[idx[rad <= thr * thr * size]] = True
    return kink


def lmi_gradient_bound(oracle, cs, x, eps, factor=4.0):
    """Per-sample yardstick for the gradient of the LMI candidate, ``[B]`` (0 where no LMI or it is not on top).

    d lambda_max / d v_a = u' G_a u with u the top eigenvector of the r x r pencil matrix S(v); an eigen-solver that
    is backward stable to ||dS|| ~ r eps ||S|| (Householder tridiagonalisation + bisection + inverse iteration, which
    is what the kernels do, and what LAPACK guarantees) returns u with an error of ||dS|| / (lambda_1 - lambda_2), and
    the gradient inherits it to first order.  The bound is ``factor * r * eps * ||S|| / (lambda_1 - lambda_2)``."""
    import torch
    buf = oracle.precompute(csd_from_cs(cs), torch.float64)
    B = x.shape[0]
    if buf["all_F"].ndim != 3:
        return np.zeros(B)
    n = cs.n
    v = x[:, 0:n, 0:1].double().cpu()
    norm = torch.linalg.vector_norm(v, dim=(1, 2)).clamp_min(1e-300)
    unit = v / norm.reshape(-1, 1, 1)
    cand = torch.nan_to_num(oracle.compute_kappa(buf, unit, terms=True), nan=0.0)
    S = torch.einsum("ajk,ial->ijk", [buf["all_F"][0:-1], buf["NA_E"] @ unit])
    lam = torch.linalg.eigvalsh(buf["L"].T @ (-S) @ buf["L"])
    r = lam.shape[1]
    if r < 2:
        return np.zeros(B)
    on_top = lam[:, -1] >= cand.max(dim=1).values * (1.0 - 1e-6)
    gap = (lam[:, -1] - lam[:, -2]).clamp_min(1e-300)
    bound = factor * r * eps * lam.abs().amax(dim=1) / gap
    return torch.where(on_top, bound, torch.zeros_like(bound)).numpy()


def residuals_device(raw, y):
    """``oracle.residuals`` and the relative form of ``test_gpu_parity._relative_violation`` for a FULL batch on the
    device: per-sample worst signed residual per family, and per-sample worst residual / (sum of |terms|), in fp64 torch
    ops on ``y``'s device (262 144 rows x 72 quadratics are minutes of numpy on the host, seconds here).  Test
    infrastructure: the callers cross-check a slice against the numpy forms of the oracle."""
    import torch
    y = y.double()
    ay = y.abs()
    dev = y.device
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64), device=dev)  # noqa: E731
    res, rel = {}, []
    if raw.get("A1") is not None:
        A, b = t(raw["A1"]), t(raw["b1"])[:, 0]
        r = y @ A.T - b
        res["lin_ineq"] = r.amax(dim=1)
        rel.append((r / (ay @ A.abs().T + b.abs())).amax(dim=1))
    if raw.get("A2") is not None:
        A, b = t(raw["A2"]), t(raw["b2"])[:, 0]
        r = (y @ A.T - b).abs()
        res["lin_eq"] = r.amax(dim=1)
        rel.append((r / (ay @ A.abs().T + b.abs() + 1e-300)).amax(dim=1))
    if len(raw.get("P", [])):
        vals, rels = [], []
        for P, q, r0 in zip(raw["P"], raw["q"], raw["r"]):
            P, q = t(P), t(q)[:, 0]
            r = 0.5 * ((y @ P) * y).sum(dim=1) + y @ q + float(r0[0, 0])
            mag = 0.5 * ((ay @ P.abs()) * ay).sum(dim=1) + ay @ q.abs() + abs(float(r0[0, 0]))
            vals.append(r)
            rels.append(r / mag)
        res["quad"] = torch.stack(vals, dim=1).amax(dim=1)
        rel.append(torch.stack(rels, dim=1).amax(dim=1))
    if len(raw.get("M", [])):
        vals, rels = [], []
        for M, s_, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"]):
            M, s_, c = t(M), t(s_)[:, 0], t(c)[:, 0]
            r = torch.linalg.norm(y @ M.T + s_, dim=1) - (y @ c + float(d[0, 0]))
            mag = torch.linalg.norm(ay @ M.abs().T + s_.abs(), dim=1) + ay @ c.abs() + abs(float(d[0, 0]))
            vals.append(r)
            rels.append(r / mag)
        res["soc"] = torch.stack(vals, dim=1).amax(dim=1)
        rel.append(torch.stack(rels, dim=1).amax(dim=1))
    if len(raw.get("F", [])):
        F = t(np.stack(raw["F"][:-1], axis=0))
        H = torch.einsum("ba,ajk->bjk", y, F) + t(raw["F"][-1])[None]
        lam = torch.linalg.eigvalsh(H)[:, 0]
        res["lmi"] = -lam
        norms = t(np.array([np.linalg.norm(Fi, 2) for Fi in raw["F"][:-1]]))
        rel.append(-lam / (ay @ norms + float(np.linalg.norm(raw["F"][-1], 2))))
    return res, torch.stack(rel, dim=1).amax(dim=1)
