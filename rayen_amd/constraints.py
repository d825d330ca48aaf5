"""Constraint description + one-time host preprocessing (numpy, fp64).

Host-side mirror of the reference's ``rayen/constraints.py``: same class names,
constructor signatures, validation rules and resulting fields, so a
``ConvexConstraints`` built for the reference can be built here unchanged and
handed to :class:`rayen_amd.constraint_module.ConstraintModule`.

* ``LinearConstraint(A1, b1, A2, b2)``          -- constraints.py:17-61
* ``ConvexQuadraticConstraint(P, q, r, ...)``   -- constraints.py:63-106
* ``SOCConstraint(M, s, c, d)``                 -- constraints.py:108-130
* ``LMIConstraint(all_F)``                      -- constraints.py:133-155
* ``ConvexConstraints(lc, qcs, socs, lmic, y0, do_preprocessing_linear,
  print_debug_info)``                           -- constraints.py:159-448

Everything here runs once, on the host, in fp64; none of it is on the per-batch
hot path.  The arithmetic that does not need an optimisation solver (stacking,
null-space reparametrisation ``y = NA_E z + yp``, ``A_p``/``b_p``, ``z0`` from a
user ``y0``) follows the reference exactly.  The reference solves its LPs and
its max-margin interior-point program with cvxpy (ECOS/SCS/Gurobi), which this
image does not ship; here those steps use HiGHS ``linprog`` for the LPs and the
small conic solver of ``rayen_amd/conic.py`` for the margin program and for
``project`` / ``getViolation``, so a *solver-chosen* ``z0`` is not bit-comparable
with the reference's ("parity unpinned").  With an
explicit ``y0`` no solver runs and the result is pinned by the golden fixtures.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg
import scipy.optimize

from . import conic, utils


def _cvxpy():
    try:
        import cvxpy as cp  # noqa: WPS433 (optional dependency)
    except ImportError as exc:  # pragma: no cover - cvxpy is absent in this image
        raise ImportError("this method needs cvxpy, which is not installed") from exc
    return cp


class LinearConstraint:
    """``A1 y <= b1`` and ``A2 y = b2``; either pair may be ``None`` (constraints.py:17-61)."""

    def __init__(self, A1, b1, A2, b2):
        self.A1 = A1
        self.b1 = b1
        self.A2 = A2
        self.b2 = b2

        utils.verify(self.hasEqConstraints() or self.hasIneqConstraints())

        if self.hasIneqConstraints():
            utils.verify(A1.ndim == 2)
            utils.verify(b1.ndim == 2)
            utils.verify(b1.shape[1] == 1)
            utils.verify(A1.shape[0] == b1.shape[0])

        if self.hasEqConstraints():
            utils.verify(A2.ndim == 2)
            utils.verify(b2.ndim == 2)
            utils.verify(b2.shape[1] == 1)
            utils.verify(A2.shape[0] == b2.shape[0])

        if self.hasIneqConstraints() and self.hasEqConstraints():
            utils.verify(A1.shape[1] == A2.shape[1])

    def hasEqConstraints(self):
        return self.A2 is not None and self.b2 is not None

    def hasIneqConstraints(self):
        return self.A1 is not None and self.b1 is not None

    def dim(self):
        if self.hasIneqConstraints():
            return self.A1.shape[1]
        return self.A2.shape[1]

    def asCvxpy(self, y, epsilon=0.0):
        constraints = []
        if self.hasIneqConstraints():
            constraints.append(self.A1 @ y <= self.b1)
        if self.hasEqConstraints():
            constraints.append(self.A2 @ y == self.b2)
        return constraints


class ConvexQuadraticConstraint:
    """``0.5 y'Py + q'y + r <= 0`` with ``P`` PSD (constraints.py:63-106)."""

    def __init__(self, P, q, r, do_checks_P=True):
        self.P = P
        self.q = q
        self.r = r

        if do_checks_P:
            utils.checkMatrixisNotZero(self.P)
            utils.checkMatrixisSymmetric(self.P)

            smallest_eigenvalue = np.amin(np.linalg.eigvalsh(self.P))
            tol = 1e-7  # PSD up to a tolerance, then lifted onto the cone (constraints.py:79-92)
            utils.verify(smallest_eigenvalue > -tol,
                         f"Matrix P is not PSD, smallest eigenvalue is {smallest_eigenvalue}")
            if -tol <= smallest_eigenvalue < 0:
                self.P = self.P + np.abs(smallest_eigenvalue) * np.eye(self.P.shape[0])

    def dim(self):
        return self.P.shape[1]

    def asCvxpy(self, y, epsilon=0.0):
        cp = _cvxpy()
        return [0.5 * cp.quad_form(y, self.P, assume_PSD=True) + self.q.T @ y + self.r <= -epsilon]


class SOCConstraint:
    """``||M y + s|| <= c'y + d`` (constraints.py:108-130)."""

    def __init__(self, M, s, c, d):
        utils.checkMatrixisNotZero(M)
        utils.checkMatrixisNotZero(c)

        utils.verify(M.shape[1] == c.shape[0])
        utils.verify(M.shape[0] == s.shape[0])
        utils.verify(s.shape[1] == 1)
        utils.verify(c.shape[1] == 1)
        utils.verify(d.shape[0] == 1)
        utils.verify(d.shape[1] == 1)

        self.M = M
        self.s = s
        self.c = c
        self.d = d

    def dim(self):
        return self.M.shape[1]

    def asCvxpy(self, y, epsilon=0.0):
        cp = _cvxpy()
        return [cp.norm(self.M @ y + self.s) - self.c.T @ y - self.d <= -epsilon]


class LMIConstraint:
    """``y_0 F_0 + ... + y_{k-1} F_{k-1} + F_k >= 0`` (constraints.py:133-155)."""

    def __init__(self, all_F):
        for F in all_F:
            utils.checkMatrixisSymmetric(F)
        for F_i in all_F:
            utils.verify(F_i.shape == all_F[0].shape)
        self.all_F = all_F

    def dim(self):
        return len(self.all_F) - 1

    def asCvxpy(self, y, epsilon=0.0):
        k = self.dim()
        lhs = 0
        for i in range(k):
            lhs += y[i, 0] * self.all_F[i]
        lhs += self.all_F[k]
        return [lhs >> epsilon * np.eye(self.all_F[0].shape[0])]


# ---------------------------------------------------------------------------
# solver-backed steps (scipy stand-ins for the reference's cvxpy programs)
# ---------------------------------------------------------------------------

def _linprog(c, A_ub, b_ub):
    """min c'z s.t. A_ub z <= b_ub, z free.  Returns (status, objective)."""
    res = scipy.optimize.linprog(c, A_ub=A_ub, b_ub=b_ub, bounds=(None, None), method="highs")
    if res.status == 0:
        return "optimal", float(res.fun)
    if res.status == 3:
        return "unbounded", -math.inf
    if res.status == 2:
        return "infeasible", math.inf
    return f"status_{res.status}", float("nan")


def _lambda_min_sym(A):
    return float(np.linalg.eigvalsh(A)[0])


class ConvexConstraints:
    """Intersection of the four constraint families plus its preprocessing (constraints.py:159-448).

    ``y0`` (a point in the relative interior) may be supplied; it is then the
    caller's responsibility that it is interior (constraints.py:160-162).
    ``do_preprocessing_linear=False`` is only legal when
    ``aff{y : A1 y <= b1} = R^k`` (constraints.py:164).
    """

    def __init__(self, lc=None, qcs=[], socs=[], lmic=None, y0=None,
                 do_preprocessing_linear=True, print_debug_info=False):

        if lc is not None:
            self.has_linear_eq_constraints = lc.hasEqConstraints()
            self.has_linear_ineq_constraints = lc.hasIneqConstraints()
            self.has_linear_constraints = (self.has_linear_eq_constraints
                                           or self.has_linear_ineq_constraints)
        else:
            self.has_linear_eq_constraints = False
            self.has_linear_ineq_constraints = False
            self.has_linear_constraints = False

        self.has_quadratic_constraints = len(qcs) > 0
        self.has_soc_constraints = len(socs) > 0
        self.has_lmi_constraints = lmic is not None

        self.lc = lc
        self.qcs = qcs
        self.socs = socs
        self.lmic = lmic

        utils.verify(self.has_quadratic_constraints or self.has_linear_constraints
                     or self.has_soc_constraints or self.has_lmi_constraints,
                     "There are no constraints!")

        all_dim = []
        if self.has_linear_constraints:
            all_dim.append(lc.dim())
        all_dim += [qc.dim() for qc in qcs]
        all_dim += [soc.dim() for soc in socs]
        if self.has_lmi_constraints:
            all_dim.append(lmic.dim())
        utils.verify(utils.all_equal(all_dim))

        self.k = all_dim[0]
        self.solver = "SCIPY"  # the reference stores the cvxpy solver name here (constraints.py:208-221)

        if y0 is None:
            # the reference checks non-emptiness with a feasibility solve (constraints.py:224-234);
            # here the interior-point search below raises the same message when it fails.
            pass

        if self.has_linear_constraints:
            A, b = self._stacked_inequalities()
            if print_debug_info:
                utils.printInBoldGreen(f"A is {A.shape} and b is {b.shape}")

            if do_preprocessing_linear:
                A, b = self._remove_redundant_rows(A, b, print_debug_info)
                E = self._equality_set(A, b, print_debug_info)
            else:
                # A_E == A2, A_I == A1 (constraints.py:331-339)
                start = self.lc.A1.shape[0] if self.has_linear_ineq_constraints else 0
                E = list(range(start, A.shape[0]))

            if print_debug_info:
                utils.printInBoldGreen(f"E={E}")
            I = [i for i in range(A.shape[0]) if i not in E]

            if len(E) > 0:
                A_E, b_E = A[E, :], b[E, :]
            else:
                A_E, b_E = np.zeros((1, A.shape[1])), np.zeros((1, 1))
            if len(I) > 0:
                A_I, b_I = A[I, :], b[I, :]
            else:
                A_I, b_I = np.zeros((1, A.shape[1])), np.ones((1, 1))  # 0 z <= 1

            # reparametrise on the null space of A_E (constraints.py:364-370)
            NA_E = scipy.linalg.null_space(A_E)
            yp = np.linalg.pinv(A_E) @ b_E
            A_p = A_I @ NA_E
            b_p = b_I - A_I @ yp

            utils.verify(A_p.ndim == 2, f"A_p.shape={A_p.shape}")
            utils.verify(b_p.ndim == 2, f"b_p.shape={b_p.shape}")
            utils.verify(b_p.shape[1] == 1)
            utils.verify(A_p.shape[0] == b_p.shape[0])
            if print_debug_info:
                utils.printInBoldGreen(f"A_p is {A_p.shape} and b_p is {b_p.shape}")
            self.n = A_p.shape[1]
        else:
            # constraints.py:383-394
            self.n = self.k
            NA_E = np.eye(self.n)
            yp = np.zeros((self.n, 1))
            A_p, b_p = np.zeros((1, self.n)), np.ones((1, 1))
            A_E, b_E = np.zeros((1, self.k)), np.zeros((1, 1))
            A_I, b_I = np.zeros((1, self.k)), np.ones((1, 1))

        self.A_E, self.b_E, self.A_I, self.b_I = A_E, b_E, A_I, b_I
        self.A_p, self.b_p, self.yp, self.NA_E = A_p, b_p, yp, NA_E

        utils.verify(self.n == (self.k - np.linalg.matrix_rank(self.A_E)))

        if y0 is None:
            self.z0 = self._find_interior_point()
            self.y0 = self.NA_E @ self.z0 + self.yp
        else:
            self.y0 = y0
            self.z0 = self.NA_E.T @ (self.y0 - self.yp)  # constraints.py:434-436

        utils.verify(np.allclose(NA_E.T @ NA_E, np.eye(NA_E.shape[1])))

    def _interior_point_with_cvxpy(self):
        """The reference's own margin program on the reference's own modelling layer, when cvxpy is importable
        ("lazy import cvxpy when installed so behaviour matches the reference exactly where it
        can"; constraints.py:412-432): maximise eps subject to every constraint of the subspace holding with margin eps,
        0 <= eps <= 0.5, handed to whichever solver cvxpy picks -- so the solver-chosen ``z0`` is the one the reference
        would have got on the same installation.  ``None`` when cvxpy is absent (this image), when
        ``RAYEN_NO_CVXPY=1``, or when its solve does not end optimal (the built-in program below then runs)."""
        import os
        if os.environ.get("RAYEN_NO_CVXPY", "0") == "1":
            return None
        try:
            import cvxpy as cp
        except Exception:               # not installed (or a broken install): the built-in solvers serve
            return None
        try:
            eps = cp.Variable()
            z = cp.Variable((self.n, 1))
            cons = self.getConstraintsInSubspaceCvxpy(z, eps) + [eps >= 0, eps <= 0.5]
            prob = cp.Problem(cp.Minimize(-eps), cons)
            prob.solve(verbose=False)
            if prob.status not in ("optimal", "optimal_inaccurate") or z.value is None or float(eps.value) <= 1e-8:
                return None
            z0 = np.asarray(z.value, dtype=np.float64).reshape(self.n, 1)
            return z0 if float(np.min(self.margins(z0))) > 1e-8 else None
        except Exception as exc:        # said, not swallowed: z0 (hence every y) then comes from the built-in program
            import warnings
            warnings.warn(f"rayen_amd: cvxpy is importable but its margin program failed ({type(exc).__name__}: {exc}); "
                          "the interior point comes from the built-in solvers instead (RAYEN_NO_CVXPY=1 skips cvxpy)",
                          UserWarning, stacklevel=2)
            return None

    # ------------------------------------------------------------------ linear preprocessing
    def _stacked_inequalities(self):
        """``A y <= b`` with equalities as two opposite inequalities (constraints.py:239-250)."""
        if self.has_linear_ineq_constraints:
            A, b = self.lc.A1, self.lc.b1
            if self.has_linear_eq_constraints:
                A = np.concatenate((A, self.lc.A2, -self.lc.A2), axis=0)
                b = np.concatenate((b, self.lc.b2, -self.lc.b2), axis=0)
        else:
            A = np.concatenate((self.lc.A2, -self.lc.A2), axis=0)
            b = np.concatenate((self.lc.b2, -self.lc.b2), axis=0)
        return A, b

    def _remove_redundant_rows(self, A, b, print_debug_info):
        """Drop row i when max A_i z s.t. the others (and A_i z <= b_i+1) stays <= b_i (constraints.py:256-286)."""
        TOL = 1e-7
        if A.shape[0] <= 1:
            return A, b
        if print_debug_info:
            utils.printInBoldBlue("Removing redundant constraints...")
        removed = 0
        for i in reversed(range(A.shape[0])):
            others = [x for x in range(A.shape[0]) if x != i]
            A_ub = np.concatenate((A[others, :], A[i:i + 1, :]), axis=0)
            b_ub = np.concatenate((b[others, 0], [b[i, 0] + 1.0]))
            status, value = _linprog(-A[i, :], A_ub, b_ub)
            if status != "optimal":
                raise Exception("Value is not optimal")
            if (-value - b[i, 0]) <= TOL:
                A = np.delete(A, i, axis=0)
                b = np.delete(b, i, axis=0)
                removed += 1
        if print_debug_info:
            utils.printInBoldBlue(f"Removed {removed} constraints ")
            utils.printInBoldGreen(f"A is {A.shape} and b is {b.shape}")
        return A, b

    def _equality_set(self, A, b, print_debug_info):
        """Rows that are tight on the whole polyhedron (constraints.py:290-329)."""
        TOL = 1e-5
        if print_debug_info:
            utils.printInBoldBlue("Finding Affine Hull and projecting...")
        E = []
        for i in range(A.shape[0]):
            status, value = _linprog(A[i, :], A, b[:, 0])
            if status == "infeasible":
                raise Exception("The feasible set is empty")
            if status not in ("optimal", "unbounded"):
                raise Exception(f"prob.status={status}")
            obj_value = value - b[i, 0] if status == "optimal" else -math.inf
            utils.verify(obj_value < TOL, f"The objective should be negative. It's {obj_value} right now")
            if obj_value > -TOL:
                E.append(i)
        return E

    # ------------------------------------------------------------------ interior point
    def margins(self, z):
        """Signed slack of every constraint at ``y = NA_E z + yp`` (positive = strictly inside).

        Same residuals as the reference's subspace program (constraints.py:499-523).
        """
        z = np.asarray(z, dtype=np.float64).reshape(self.n, 1)
        y = self.NA_E @ z + self.yp
        out = [(self.b_p - self.A_p @ z).ravel()]
        for qc in self.qcs:
            out.append(-(0.5 * y.T @ qc.P @ y + qc.q.T @ y + qc.r).ravel())
        for soc in self.socs:
            out.append(((soc.c.T @ y + soc.d) - np.linalg.norm(soc.M @ y + soc.s)).ravel())
        if self.has_lmi_constraints:
            H = self.lmic.all_F[-1].astype(np.float64).copy()
            for i in range(self.lmic.dim()):
                H = H + y[i, 0] * self.lmic.all_F[i]
            out.append(np.array([_lambda_min_sym(H)]))
        return np.concatenate(out)

    def _nonlinear_cone_rows(self, prog, Y, y_off, eps_col=None):
        """Append the quadratic / SOC / LMI constraints at ``y = Y x + y_off`` to ``prog`` (a
        :class:`conic.ConeProgram`), each tightened by the margin ``x[eps_col]`` when given --
        the conic form of ``asCvxpy(y, epsilon)`` (constraints.py:104-106, 129-130, 147-155)."""
        nvar = prog.nvar
        e = np.zeros(nvar)
        if eps_col is not None:
            e[eps_col] = 1.0
        y_off = y_off.reshape(-1)
        for qc in self.qcs:
            q = qc.q.reshape(-1)
            # 1/2 y'Py <= t,  t = -q'y - r - eps
            prog.add_quadratic(qc.P, (Y, y_off), (-(q @ Y) - e, -float(q @ y_off) - float(qc.r.item())))
        for soc in self.socs:
            cvec = soc.c.reshape(-1)
            G = np.concatenate((soc.M @ Y, ((cvec @ Y) - e).reshape(1, -1)), axis=0)
            h = np.concatenate((soc.M @ y_off + soc.s.reshape(-1), [float(cvec @ y_off) + float(soc.d.item())]))
            prog.add(conic.SOC, G, h)
        if self.has_lmi_constraints:
            F = [np.asarray(Fi, dtype=np.float64) for Fi in self.lmic.all_F]
            r = F[0].shape[0]
            Fy = np.stack([Fi.reshape(-1) for Fi in F[:-1]], axis=1)          # [r*r, k]
            G = Fy @ Y - np.outer(np.eye(r).reshape(-1), e)
            h = Fy @ y_off + F[-1].reshape(-1)
            prog.add(conic.PSD, G, h, dim=r)

    def _find_interior_point(self):
        """max eps s.t. every constraint holds with margin eps, 0<=eps<=0.5 (constraints.py:412-432).

        Linear-only sets are an LP (HiGHS).  With nonlinear families present the same
        margin program is solved as a conic program (``rayen_amd/conic.py``: quadratics and
        cones as second-order cones, the LMI as a PSD cone -- no derivative of ``lambda_min``
        is needed, so structured LMIs with repeated eigenvalues are handled); the result
        only has to be strictly interior, which is verified on the true residuals.
        """
        n = self.n
        m = self.A_p.shape[0]
        z_cvx = self._interior_point_with_cvxpy()
        if z_cvx is not None:
            return z_cvx
        # LP over (z, eps): A_p z + eps <= b_p
        c = np.zeros(n + 1)
        c[-1] = -1.0
        A_ub = np.concatenate((self.A_p, np.ones((m, 1))), axis=1)
        bounds = [(None, None)] * n + [(0.0, 0.5)]
        res = scipy.optimize.linprog(c, A_ub=A_ub, b_ub=self.b_p[:, 0], bounds=bounds, method="highs")
        nonlinear = (self.has_quadratic_constraints or self.has_soc_constraints
                     or self.has_lmi_constraints)
        if res.status == 2:
            raise Exception("The feasible set is empty")
        if res.status == 0:
            z_start, eps_lp = res.x[:n], res.x[-1]
        else:  # unbounded in z for a fixed eps cannot happen (eps bounded); be defensive
            z_start, eps_lp = np.zeros(n), 0.0
        if not nonlinear:
            utils.verify(eps_lp > 1e-8)
            return z_start.reshape(n, 1)

        prog = conic.ConeProgram(n + 1)
        prog.add(conic.NONNEG, -A_ub, self.b_p[:, 0])                       # b_p - A_p z - eps >= 0
        box = np.zeros((2, n + 1))
        box[0, -1], box[1, -1] = 1.0, -1.0
        prog.add(conic.NONNEG, box, [0.0, 0.5])                             # 0 <= eps <= 0.5
        Y = np.concatenate((self.NA_E, np.zeros((self.k, 1))), axis=1)
        self._nonlinear_cone_rows(prog, Y, self.yp, eps_col=n)
        x, info = conic.solve(prog, None, c, x0=np.concatenate((z_start, [0.0])), eps_abs=1e-8, eps_rel=1e-8)
        z = x[:n]
        eps = float(np.min(self.margins(z)))
        if eps <= 1e-8:
            # the margin program did not deliver a strictly interior point: the set has an empty interior
            # in the subspace (or no point at all) -- same outcome as constraints.py:224-234 / :428
            raise Exception("The feasible set is empty")
        return z.reshape(n, 1)

    # ------------------------------------------------------------------ export
    def getDataAsDict(self):
        """Constraint data with neutral fillers for absent families (constraints.py:451-494)."""
        if self.has_linear_eq_constraints:
            A2, b2 = self.lc.A2, self.lc.b2
        else:
            A2, b2 = np.zeros((1, self.k)), np.array([[0]])
        if self.has_linear_ineq_constraints:
            A1, b1 = self.lc.A1, self.lc.b1
        else:
            A1, b1 = np.zeros((1, self.k)), np.array([[1]])
        if self.has_quadratic_constraints:
            all_P, all_q, all_r = utils.getAllPqrFromQcs(self.qcs)
        else:
            all_P, all_q, all_r = [np.zeros((self.k, self.k))], [np.zeros((self.k, 1))], [-np.ones((1, 1))]
        if self.has_soc_constraints:
            all_M, all_s, all_c, all_d = utils.getAllMscdFromSocs(self.socs)
        else:
            all_M, all_s = [np.zeros((self.k, self.k))], [np.zeros((self.k, 1))]
            all_c, all_d = [np.zeros((self.k, 1))], [np.ones((1, 1))]
        if self.has_lmi_constraints:
            all_F = self.lmic.all_F
        else:
            all_F = [np.zeros((self.k, self.k)) for _ in range(self.k)] + [np.eye(self.k)]
        return dict(A2=A2, b2=b2, A1=A1, b1=b1, all_P=all_P, all_q=all_q, all_r=all_r,
                    all_M=all_M, all_s=all_s, all_c=all_c, all_d=all_d, all_F=all_F)

    # ------------------------------------------------------------------ residual metric
    def getResiduals(self, y):
        """Worst signed residual per family for a batch ``y [B,k]`` (fp64; positive = violated).

        The reference's ``getViolation`` (constraints.py:549-559) is a cvxpy
        projection; the quantities below are the ones its soft cost uses
        (examples/cost_computer.py:69-110) and are what SURVEY.md §8(d) defines as
        the violation metric of this build.
        """
        y = np.asarray(y, dtype=np.float64).reshape(-1, self.k)
        res = {}
        if self.has_linear_ineq_constraints:
            res["lin_ineq"] = np.max(y @ self.lc.A1.T - self.lc.b1.T, axis=1)
        if self.has_linear_eq_constraints:
            res["lin_eq"] = np.max(np.abs(y @ self.lc.A2.T - self.lc.b2.T), axis=1)
        if self.has_quadratic_constraints:
            vals = [0.5 * np.einsum("bi,ij,bj->b", y, qc.P, y) + y @ qc.q[:, 0] + qc.r[0, 0]
                    for qc in self.qcs]
            res["quad"] = np.max(np.stack(vals, axis=1), axis=1)
        if self.has_soc_constraints:
            vals = [np.linalg.norm(y @ soc.M.T + soc.s.T, axis=1) - (y @ soc.c[:, 0] + soc.d[0, 0])
                    for soc in self.socs]
            res["soc"] = np.max(np.stack(vals, axis=1), axis=1)
        if self.has_lmi_constraints:
            F = np.stack(self.lmic.all_F[:-1], axis=0)
            H = np.einsum("ba,ajk->bjk", y, F) + self.lmic.all_F[-1][None]
            res["lmi"] = -np.linalg.eigvalsh(H)[:, 0]
        return res

    def getViolationRows(self, y):
        """Per sample: its largest residual over all families, ``[B]`` (<=0 = that sample is feasible)."""
        res = self.getResiduals(y)
        return np.max(np.stack(list(res.values()), axis=0), axis=0)

    def getMaxViolation(self, y):
        """Largest residual over all families and samples (<=0 means every sample is feasible)."""
        res = self.getResiduals(y)
        return max(float(np.max(v)) for v in res.values())

    # ------------------------------------------------------------------ cvxpy-only surface
    def getConstraintsInSubspaceCvxpy(self, z, epsilon=0.0):
        constraints = [self.A_p @ z - self.b_p <= -epsilon * np.ones((self.A_p.shape[0], 1))]
        y = self.NA_E @ z + self.yp
        return constraints + self.getNonLinearConstraintsCvxpy(y, epsilon)

    def getNonLinearConstraintsCvxpy(self, y, epsilon=0.0):
        constraints = []
        for qc in self.qcs:
            constraints += qc.asCvxpy(y, epsilon)
        for soc in self.socs:
            constraints += soc.asCvxpy(y, epsilon)
        if self.has_lmi_constraints:
            constraints += self.lmic.asCvxpy(y, epsilon)
        return constraints

    def getConstraintsCvxpy(self, y, epsilon=0.0):
        constraints = []
        if self.has_linear_constraints:
            constraints += self.lc.asCvxpy(y, epsilon)
        return constraints + self.getNonLinearConstraintsCvxpy(y, epsilon)

    def project(self, y_to_be_projected):
        """Euclidean projection onto the set (constraints.py:443-447, 539-547): ``(y_projected [k,1],
        squared distance)``.  The reference solves this program with cvxpy; here it is the same
        program in conic form on ``rayen_amd/conic.py`` (no cvxpy in this image)."""
        p = np.asarray(y_to_be_projected, dtype=np.float64).reshape(self.k)
        k = self.k
        prog = conic.ConeProgram(k)
        if self.has_linear_ineq_constraints:
            prog.add(conic.NONNEG, -self.lc.A1, self.lc.b1[:, 0])
        if self.has_linear_eq_constraints:
            prog.add(conic.ZERO, self.lc.A2, -self.lc.b2[:, 0])
        self._nonlinear_cone_rows(prog, np.eye(k), np.zeros(k))
        # min ||y - p||^2 = y'y - 2p'y + p'p  ->  P = 2I, c = -2p
        y, info = conic.solve(prog, 2.0 * np.eye(k), -2.0 * p, x0=p)
        if info["status"] != "solved" and info["r_prim"] > 1e-6:
            raise Exception(f"Value is not optimal, prob_status={info['status']}")
        return y.reshape(k, 1), float(np.sum((y - p) ** 2))

    def getViolation(self, y_to_be_projected):
        """Squared distance of a point to the set (constraints.py:549-559); 0 for feasible points."""
        if y_to_be_projected.ndim == 1:
            y_to_be_projected = np.expand_dims(y_to_be_projected, axis=1)
        _, violation = self.project(y_to_be_projected)
        return violation
