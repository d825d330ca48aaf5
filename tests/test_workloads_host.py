"""Host-side checks of rayen_amd/workloads.py (no GPU): the violation report of bench.py and the corridor set's counts."""
import numpy as np

from rayen_amd import workloads


def test_violation_report_reads_residuals_against_fp32_rounding():
    cs = workloads.build_constraints(workloads.make_raw("c3", seed=0))
    y0 = cs.y0[:, 0][None, :]
    inside = workloads.violation_report(cs, y0)
    assert all(f["max_residual"] < 0 and f["over_f32_rounding_of_y"] == 0.0 for f in inside["per_family"].values())
    # a point pushed one unit along a face normal violates that linear row by the row's norm squared, far above rounding
    a = cs.lc.A1[0]
    t = (cs.lc.b1[0, 0] - a @ y0[0]) / (a @ a)
    out = workloads.violation_report(cs, y0 + (t + 1e-3) * a[None, :])
    assert out["per_family"]["lin_ineq"]["max_residual"] > 0
    assert out["per_family"]["lin_ineq"]["over_f32_rounding_of_y"] > 100.0
    # a feasible point rounded to fp32 sits within a few units of the rounding yardstick
    on_face = (y0 + t * a[None, :]).astype(np.float32)
    rep = workloads.violation_report(cs, on_face)
    assert rep["per_family"]["lin_ineq"]["over_f32_rounding_of_y"] <= 8.0


def test_corridor_set_has_the_generators_counts():
    """rayen_amd/workloads.py::corridor_spline against the counts SURVEY.md 8(d) derives from the MATLAB generator:
    k = 45 (15 control points in 3-D), 15 equalities (n = 30), 72 quadratic limits, ~10^3 face rows."""
    raw = workloads.make_raw("c5", seed=0)
    assert raw["A2"].shape == (15, 45) and len(raw["P"]) == 72
    assert 900 <= raw["A1"].shape[0] <= 1300
    cs = workloads.build_constraints(raw)
    assert cs.k == 45 and cs.n == 30
    assert cs.getMaxViolation(cs.y0[:, 0][None, :]) < 0 or abs(cs.getMaxViolation(cs.y0[:, 0][None, :])) < 1e-9
    ranks = [np.linalg.matrix_rank(P) for P in raw["P"]]
    assert max(ranks) == 3
