"""A small conic-QP solver for the set-up steps that the reference hands to cvxpy.

The reference solves three kinds of programs off the hot path with cvxpy + a conic
solver (ECOS / SCS / Gurobi, none of which is in this image):

* the max-margin interior point ``max eps`` subject to every constraint holding
  with margin ``eps`` (rayen/constraints.py:412-432),
* the feasibility check of the set (constraints.py:224-234),
* the Euclidean projection behind ``project`` / ``getViolation`` (constraints.py:443-447, 539-559).

All three are ``min 1/2 x'Px + c'x  s.t.  G x + h in K`` with ``K`` a product of
non-negative orthants, zero cones, second-order cones and PSD cones.  This module
solves that form with the operator-splitting iteration of OSQP / COSMO (ADMM on
``G x = z, z in K - h`` with over-relaxation and residual-balanced ``rho``): the
problems are tiny (tens of variables) and run once per constraint set, on the host,
in fp64 numpy.  Nothing here is on the per-batch path.

Convex quadratic constraints enter as second-order cones: with ``P = T T'``,
``1/2 y'Py + q'y + r <= -eps``  <=>  ``||(sqrt2 T'y, t - 1)|| <= t + 1`` for
``t = -q'y - r - eps`` (both sides squared: ``2||T'y||^2 <= 4t``).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg

NONNEG, ZERO, SOC, PSD = "nonneg", "zero", "soc", "psd"


def _proj_cone(kind, w, dim):
    if kind == NONNEG:
        return np.maximum(w, 0.0)
    if kind == ZERO:
        return np.zeros_like(w)
    if kind == SOC:  # (x, t) with ||x|| <= t, t stored LAST
        x, t = w[:-1], w[-1]
        nx = float(np.linalg.norm(x))
        if nx <= t:
            return w.copy()
        if nx <= -t:
            return np.zeros_like(w)
        a = 0.5 * (nx + t)
        out = np.empty_like(w)
        out[:-1] = a * x / nx
        out[-1] = a
        return out
    if kind == PSD:  # full r x r storage, row-major; Frobenius norm = Euclidean norm of the block
        M = w.reshape(dim, dim)
        M = 0.5 * (M + M.T)
        lam, V = np.linalg.eigh(M)
        return ((V * np.maximum(lam, 0.0)) @ V.T).reshape(-1)
    raise ValueError(kind)


class ConeProgram:
    """Accumulates the rows of ``G x + h in K`` block by block."""

    def __init__(self, nvar):
        self.nvar = nvar
        self.G, self.h, self.cones = [], [], []   # cones: (kind, rows, dim)

    def add(self, kind, G, h, dim=0):
        G = np.asarray(G, dtype=np.float64).reshape(-1, self.nvar)
        h = np.asarray(h, dtype=np.float64).reshape(-1)
        assert G.shape[0] == h.shape[0]
        if G.shape[0]:
            self.G.append(G)
            self.h.append(h)
            self.cones.append((kind, G.shape[0], dim))

    def add_quadratic(self, P, q_aff, t_aff):
        """``1/2 u'Pu <= t`` where ``u = Uy x + u0`` and ``t = t_aff[0] x + t_aff[1]`` are affine in x.

        ``q_aff = (Uy, u0)``.  P must be PSD (checked by the constraint classes)."""
        Uy, u0 = q_aff
        tg, t0 = t_aff
        lam, V = np.linalg.eigh(0.5 * (P + P.T))
        keep = lam > 1e-14 * max(float(lam[-1]), 1e-300)
        Tt = (np.sqrt(lam[keep])[:, None]) * V[:, keep].T            # T' with P = T T'
        G = np.concatenate((np.sqrt(2.0) * Tt @ Uy, tg.reshape(1, -1), tg.reshape(1, -1)), axis=0)
        h = np.concatenate((np.sqrt(2.0) * Tt @ u0, [t0 - 1.0], [t0 + 1.0]))
        self.add(SOC, G, h)

    def stacked(self):
        if not self.G:
            return np.zeros((0, self.nvar)), np.zeros(0)
        return np.concatenate(self.G, axis=0), np.concatenate(self.h)

    def project(self, w):
        out = np.empty_like(w)
        at = 0
        for kind, rows, dim in self.cones:
            out[at:at + rows] = _proj_cone(kind, w[at:at + rows], dim)
            at += rows
        return out


def solve(prog: ConeProgram, P, c, x0=None, max_iter=20000, eps_abs=1e-9, eps_rel=1e-9):
    """``min 1/2 x'Px + c'x  s.t.  G x + h in K``.  Returns ``(x, info)`` with
    ``info = {"status": "solved" | "max_iter", "iters", "r_prim", "r_dual", "objective"}``.

    A problem without a feasible point does not converge: the caller reads ``r_prim``."""
    n = prog.nvar
    G, h = prog.stacked()
    m = G.shape[0]
    P = np.zeros((n, n)) if P is None else np.asarray(P, dtype=np.float64)
    c = np.asarray(c, dtype=np.float64).reshape(n)
    # equilibrate the rows of G block-wise (one scale per cone keeps every cone a cone)
    scale = np.ones(m)
    at = 0
    for kind, rows, dim in prog.cones:
        nrm = float(np.linalg.norm(np.concatenate((G[at:at + rows], h[at:at + rows, None]), axis=1))) / np.sqrt(rows)
        if kind == NONNEG or kind == ZERO:   # rows of an orthant may be scaled one by one
            rn = np.linalg.norm(np.concatenate((G[at:at + rows], h[at:at + rows, None]), axis=1), axis=1)
            scale[at:at + rows] = 1.0 / np.where(rn > 0, rn, 1.0)
        elif nrm > 0:
            scale[at:at + rows] = 1.0 / nrm
        at += rows
    Gs, hs = G * scale[:, None], h * scale
    sigma, alpha, rho = 1e-6, 1.6, 1.0
    x = np.zeros(n) if x0 is None else np.asarray(x0, dtype=np.float64).reshape(n).copy()
    z = prog.project(Gs @ x + hs) - hs
    y = np.zeros(m)

    def factor(rho):
        K = P + sigma * np.eye(n) + rho * (Gs.T @ Gs)
        return scipy.linalg.cho_factor(K)

    fac = factor(rho)
    status, it = "max_iter", 0
    r_prim = r_dual = np.inf
    for it in range(1, max_iter + 1):
        xt = scipy.linalg.cho_solve(fac, sigma * x - c + Gs.T @ (rho * z - y))
        zt = Gs @ xt
        x = alpha * xt + (1.0 - alpha) * x
        zr = alpha * zt + (1.0 - alpha) * z
        z_new = prog.project(zr + y / rho + hs) - hs
        y = y + rho * (zr - z_new)
        z = z_new
        if it % 10 == 0 or it == max_iter:
            Gx = Gs @ x
            r_prim = float(np.max(np.abs(Gx - z))) if m else 0.0
            Px = P @ x
            Gty = Gs.T @ y
            r_dual = float(np.max(np.abs(Px + c + Gty))) if n else 0.0
            tol_p = eps_abs + eps_rel * max(float(np.max(np.abs(Gx), initial=0.0)), float(np.max(np.abs(z), initial=0.0)))
            tol_d = eps_abs + eps_rel * max(float(np.max(np.abs(Px), initial=0.0)), float(np.max(np.abs(c), initial=0.0)),
                                            float(np.max(np.abs(Gty), initial=0.0)))
            if r_prim <= tol_p and r_dual <= tol_d:
                status = "solved"
                break
            if it % 100 == 0:   # residual balancing
                num = r_prim / max(float(np.max(np.abs(Gx), initial=0.0)), float(np.max(np.abs(z), initial=0.0)), 1e-12)
                den = r_dual / max(float(np.max(np.abs(Px), initial=0.0)), float(np.max(np.abs(c), initial=0.0)),
                                   float(np.max(np.abs(Gty), initial=0.0)), 1e-12)
                new_rho = float(np.clip(rho * np.sqrt(max(num, 1e-12) / max(den, 1e-12)), 1e-6, 1e6))
                if new_rho > 5.0 * rho or new_rho < 0.2 * rho:
                    rho = new_rho
                    fac = factor(rho)
    info = {"status": status, "iters": it, "r_prim": r_prim, "r_dual": r_dual,
            "objective": float(0.5 * x @ P @ x + c @ x)}
    return x, info
