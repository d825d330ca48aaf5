"""CPU oracle for RAYEN's ``ConstraintModule.forward`` (method ``'RAYEN'``).

TEST INFRASTRUCTURE ONLY.  Nothing under ``rayen_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may.  It is the checker (and the timed CPU baseline), never the product.

It is a plain PyTorch-CPU restatement of the reference's algorithm that issues
the *same op sequence* as ``/root/reference/rayen/constraint_module.py`` so that
(i) it can be pinned against the real reference (``tests/golden/*.npz``, produced
by ``tests/golden/make_golden.py`` which imports the reference in the build
container) and (ii) timing it on the GPU box's host cores is a fair stand-in for
the reference's own PyTorch-CPU path (``cpu_baseline.kind = "port"``).

Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks this file against
outputs of the reference itself on the 15 example sets of
``examples/examples_sets.py`` and on B=256 slices of BASELINE.json's configs 2-5
(fp32 and fp64).

Function -> reference lines it follows (CM = rayen/constraint_module.py):

* ``precompute``      CM:38 (D), CM:43-52 (H, L for the LMI), CM:59-74 (casts to
                      the torch default dtype), CM:99-122 (sigma, phi, delta)
* ``solve_second_order`` CM:339-348
* ``compute_kappa``   CM:351-458
* ``forward_rayen``   CM:468-474 with ``getyFromz`` CM:512-514
* ``forward_rayen_old`` CM:460-466
* ``forward``         CM:520-533 with ``create_map=False`` (identity mapper)
"""
from __future__ import annotations

import numpy as np
import torch


def precompute(cs: dict, dtype=torch.float32) -> dict:
    """Constructor-time constants of the layer, as tensors of ``dtype``.

    ``cs`` holds the fp64 numpy fields the reference reads from its
    ``ConvexConstraints`` object: ``A_p, b_p, NA_E, yp, z0, y0`` plus the lists
    ``P, q, r`` (quadratic), ``M, s, c, d`` (SOC) and ``F`` (LMI, k+1 matrices).
    """
    def cast(a):
        return torch.tensor(np.asarray(a, dtype=np.float64)).to(dtype)

    n = cs["A_p"].shape[1]
    # CM:38 -- every row of A_p scaled by its slack at z0 (numpy fp64, then cast)
    slack = cs["b_p"] - cs["A_p"] @ cs["z0"]
    buf = {"D": cast(cs["A_p"] / (slack @ np.ones((1, n))))}
    for name in ("A_p", "b_p", "yp", "NA_E", "z0", "y0"):
        buf[name] = cast(cs[name])

    P, q, r = cs.get("P", []), cs.get("q", []), cs.get("r", [])
    M, s, c, d = cs.get("M", []), cs.get("s", []), cs.get("c", []), cs.get("d", [])
    F = cs.get("F", [])
    buf["all_P"] = cast(np.array(P)) if len(P) else torch.zeros(0, dtype=dtype)
    buf["all_q"] = cast(np.array(q)) if len(q) else torch.zeros(0, dtype=dtype)
    buf["all_r"] = cast(np.array(r)) if len(r) else torch.zeros(0, dtype=dtype)
    buf["all_M"] = cast(np.array(M)) if len(M) else torch.zeros(0, dtype=dtype)
    buf["all_s"] = cast(np.array(s)) if len(s) else torch.zeros(0, dtype=dtype)
    buf["all_c"] = cast(np.array(c)) if len(c) else torch.zeros(0, dtype=dtype)
    buf["all_d"] = cast(np.array(d)) if len(d) else torch.zeros(0, dtype=dtype)

    if len(F):
        # CM:43-52 -- H = F_k + sum_i y0_i F_i ; H^-1 = L L' (numpy fp64)
        H = np.array(F[-1], dtype=np.float64).copy()
        for i in range(len(F) - 1):
            H = H + cs["y0"][i, 0] * np.asarray(F[i], dtype=np.float64)
        L = np.linalg.cholesky(np.linalg.inv(H))
        buf["L"] = cast(L)
        buf["all_F"] = cast(np.array(list(F[:-1]) + [H]))  # CM:44-47 leaves H in the last slot
    else:
        buf["all_F"] = torch.zeros(0, dtype=dtype)

    if len(P):
        # CM:105-119 -- evaluated in torch at ``dtype`` from the already-cast buffers
        y0 = buf["y0"]
        phis, deltas = [], []
        for i in range(buf["all_P"].shape[0]):
            Pi, qi, ri = buf["all_P"][i], buf["all_q"][i], buf["all_r"][i]
            g_y0 = 0.5 * y0.T @ Pi @ y0 + qi.T @ y0 + ri
            sigma = 2 * g_y0
            grad_row = y0.T @ Pi + qi.T
            phis.append(-grad_row / sigma)
            deltas.append((grad_row.T @ grad_row - 4 * g_y0 * 0.5 * Pi) / torch.square(sigma))
        buf["all_phi"] = torch.stack(phis)
        buf["all_delta"] = torch.stack(deltas)
    return buf


def solve_second_order(a, b, c):
    """Largest non-negative root of ``a x^2 + b x + c = 0`` (CM:339-348, SOC branch)."""
    disc = torch.square(b) - 4 * a * c
    root = torch.sqrt(disc)
    sol1 = (-b - root) / (2 * a)
    sol2 = (-b + root) / (2 * a)
    return torch.relu(torch.maximum(sol1, sol2))


def compute_kappa(buf: dict, v_bar: torch.Tensor, terms: bool = False) -> torch.Tensor:
    """``kappa [B,1,1]`` for directions ``v_bar [B,n,1]`` (CM:351-458).

    ``terms=True`` (test helper, not in the reference): return instead every candidate that enters the final
    maxima -- ``[B, m + Q + S + 2]``: each linear row, each quadratic, each cone, and the two largest LMI
    eigenvalues -- so that a test can tell samples sitting on a kink of kappa (two candidates tie) apart."""
    lin = buf["D"] @ v_bar
    kappa = torch.relu(torch.max(lin, dim=1, keepdim=True).values)  # CM:353
    lam_top2 = None

    n_quad = buf["all_P"].shape[0] if buf["all_P"].ndim == 3 else 0
    n_soc = buf["all_M"].shape[0] if buf["all_M"].ndim == 3 else 0
    has_lmi = buf["all_F"].ndim == 3
    if n_quad or n_soc or has_lmi:
        rho = buf["NA_E"] @ v_bar                       # CM:356
        rhoT = torch.transpose(rho, 1, 2)
        parts = torch.empty((v_bar.shape[0], 0, 1), dtype=v_bar.dtype)

        for i in range(n_quad):                         # CM:360-381
            k_i = buf["all_phi"][i] @ rho + torch.sqrt(rhoT @ buf["all_delta"][i] @ rho)
            parts = torch.cat((parts, k_i), dim=1)

        for j in range(n_soc):                          # CM:383-399
            M, s, c, d = buf["all_M"][j], buf["all_s"][j], buf["all_c"][j], buf["all_d"][j]
            beta = M @ buf["y0"] + s
            tau = c.T @ buf["y0"] + d
            c_p = rhoT @ M.T @ M @ rho - torch.square(c.T @ rho)
            b_p = 2 * rhoT @ M.T @ beta - 2 * (c.T @ rho) @ tau
            a_p = beta.T @ beta - torch.square(tau)
            parts = torch.cat((parts, solve_second_order(a_p, b_p, c_p)), dim=1)

        if has_lmi:                                     # CM:401-449
            S = torch.einsum("ajk,ial->ijk", [buf["all_F"][0:-1], rho])
            sym = buf["L"].T @ (-S) @ buf["L"]
            lam = torch.linalg.eigvalsh(sym).unsqueeze(2)
            lam_top2 = lam[:, -2:, :] if lam.shape[1] >= 2 else torch.cat((lam, lam), dim=1)
            parts = torch.cat((parts, torch.relu(torch.max(lam, dim=1, keepdim=True).values)), dim=1)

        if terms:
            cand = [lin, parts[:, :-1] if has_lmi else parts]
            if has_lmi:
                cand.append(lam_top2)
            return torch.cat(cand, dim=1)[:, :, 0]
        kappa = torch.maximum(kappa, torch.max(parts, dim=1, keepdim=True).values)  # CM:452-453
    if terms:
        return lin[:, :, 0]
    return kappa


def forward_rayen(buf: dict, q: torch.Tensor) -> torch.Tensor:
    """``q [B, >=n, 1] -> y [B,k,1]`` (CM:468-474, 512-514)."""
    n = buf["NA_E"].shape[1]
    v = q[:, 0:n, 0:1]
    v_bar = torch.nn.functional.normalize(v, dim=1)
    kappa = compute_kappa(buf, v_bar)
    norm_v = torch.linalg.vector_norm(v, dim=(1, 2), keepdim=True)
    alpha = torch.minimum(1 / kappa, norm_v)
    return buf["NA_E"] @ (buf["z0"] + alpha * v_bar) + buf["yp"]


def forward_rayen_old(buf: dict, q: torch.Tensor) -> torch.Tensor:
    """``q [B, >=n+1, 1] -> y [B,k,1]``: the ``RAYEN_old`` head, step ``1/(exp(beta)+kappa)`` (CM:460-466)."""
    n = buf["NA_E"].shape[1]
    v = q[:, 0:n, 0:1]
    v_bar = torch.nn.functional.normalize(v, dim=1)
    kappa = compute_kappa(buf, v_bar)
    beta = q[:, n:(n + 1), 0:1]
    alpha = 1 / (torch.exp(beta) + kappa)
    return buf["NA_E"] @ (buf["z0"] + alpha * v_bar) + buf["yp"]


def forward(buf: dict, x: torch.Tensor, method: str = "RAYEN", check_nan: bool = True) -> torch.Tensor:
    """Layer forward with the identity mapper (CM:520-533, ``create_map=False``).

    ``check_nan=False`` = the reference under ``python -O`` (CM:531's assert compiled out): used only where the
    reference's own fp32 arithmetic is known to produce NaN (tests/golden/config_c5s.npz, helpers.load_golden)."""
    q = torch.flatten(x, 1).unsqueeze(2)  # == x.view(B, -1), and defined for B = 0
    y = forward_rayen(buf, q) if method == "RAYEN" else forward_rayen_old(buf, q)
    if check_nan:
        assert not torch.isnan(y).any()
    return y


# ---------------------------------------------------------------------------
# constraint residuals, fp64 numpy (SURVEY.md §8d violation metric; the same
# quantities examples/cost_computer.py:69-110 penalises)
# ---------------------------------------------------------------------------

def residuals(raw: dict, y) -> dict:
    """Worst signed residual per family for ``y [B,k]``; positive means violated.

    ``raw`` holds the user-level data: optional ``A1,b1,A2,b2`` and lists
    ``P,q,r``, ``M,s,c,d``, ``F``.
    """
    y = np.asarray(y, dtype=np.float64).reshape(y.shape[0], -1)
    out = {}
    if raw.get("A1") is not None:
        out["lin_ineq"] = np.max(y @ raw["A1"].T - raw["b1"].T, axis=1)
    if raw.get("A2") is not None:
        out["lin_eq"] = np.max(np.abs(y @ raw["A2"].T - raw["b2"].T), axis=1)
    if len(raw.get("P", [])):
        vals = [0.5 * np.einsum("bi,ij,bj->b", y, P, y) + y @ q[:, 0] + r[0, 0]
                for P, q, r in zip(raw["P"], raw["q"], raw["r"])]
        out["quad"] = np.max(np.stack(vals, axis=1), axis=1)
    if len(raw.get("M", [])):
        vals = [np.linalg.norm(y @ M.T + s.T, axis=1) - (y @ c[:, 0] + d[0, 0])
                for M, s, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"])]
        out["soc"] = np.max(np.stack(vals, axis=1), axis=1)
    if len(raw.get("F", [])):
        F = np.stack(raw["F"][:-1], axis=0)
        H = np.einsum("ba,ajk->bjk", y, F) + np.asarray(raw["F"][-1])[None]
        out["lmi"] = -np.linalg.eigvalsh(H)[:, 0]
    return out


def max_violation(raw: dict, y) -> float:
    return max(float(np.max(v)) for v in residuals(raw, y).values())
