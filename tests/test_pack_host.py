"""Host logic without a GPU: constraint preprocessing, module buffers, constant packing."""
import numpy as np
import pytest
import torch

from helpers import golden_names, load_golden, rel_err_rows, csd_from_cs
from oracle import rayen_oracle as oracle
from packed_eval import evaluate
from rayen_amd import _lib, workloads
from rayen_amd.constraint_module import ConstraintModule
from rayen_amd.pack import pack_constants


def _module_from_raw(raw, dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(raw)
        return cs, ConstraintModule(cs, method="RAYEN", create_map=False)
    finally:
        torch.set_default_dtype(prev)


def _to_my_basis(cs, csd_ref, x):
    """Golden inputs are coordinates in the reference's null-space basis; re-express them in ours."""
    R = cs.NA_E.T @ csd_ref["NA_E"]                       # orthogonal n x n
    return (x[:, :, 0].astype(np.float64) @ R.T)


@pytest.mark.parametrize("name", golden_names())
def test_preprocessing_matches_reference(name):
    """ConvexConstraints with explicit y0: same subspace, same A_p/b_p/yp/z0 up to the basis choice."""
    raw, csd, z = load_golden(name)
    cs = workloads.build_constraints(raw)
    assert (cs.k, cs.n) == (csd["NA_E"].shape[0], csd["NA_E"].shape[1])
    # same subspace: projectors agree
    assert np.allclose(cs.NA_E @ cs.NA_E.T, csd["NA_E"] @ csd["NA_E"].T, atol=1e-12)
    assert np.allclose(cs.yp, csd["yp"], atol=1e-12)
    assert np.allclose(cs.NA_E @ cs.z0, csd["NA_E"] @ csd["z0"], atol=1e-12)
    assert np.allclose(cs.b_p, csd["b_p"], atol=1e-12)
    R = cs.NA_E.T @ csd["NA_E"]
    assert np.allclose(cs.A_p @ R, csd["A_p"], atol=1e-12)


@pytest.mark.parametrize("name", golden_names())
def test_module_buffers_match_reference(name):
    """Buffers the reference registers (D, all_phi, all_delta, L) come out the same at fp32 and fp64."""
    raw, csd, z = load_golden(name)
    for tag, dtype, tol in (("64", torch.float64, 1e-12), ("32", torch.float32, 2e-6)):
        cs, layer = _module_from_raw(raw, dtype)
        if np.allclose(cs.NA_E, csd["NA_E"]):
            ref = z["buf_D" + tag]
            assert np.max(np.abs(layer.D.numpy() - ref)) <= tol * max(1.0, np.max(np.abs(ref)))
        for bname in ("all_phi", "all_delta", "L"):
            key = f"buf_{bname}{tag}"
            if key in z:
                ref = z[key]
                got = getattr(layer, bname).numpy()
                assert got.shape == ref.shape
                assert np.max(np.abs(got - ref)) <= tol * max(1.0, np.max(np.abs(ref))), bname
        assert layer.getDimAfterMap() == cs.n
        sd = layer.state_dict()
        for key in ("D", "all_P", "all_q", "all_r", "all_M", "all_s", "all_c", "all_d", "all_F",
                    "A_p", "b_p", "yp", "NA_E", "z0", "y0"):
            assert key in sd


@pytest.mark.parametrize("name", golden_names())
def test_packed_form_reproduces_reference_fp64(name):
    """numpy evaluation of (W, segments) == the reference's fp64 forward on the golden inputs."""
    raw, csd, z = load_golden(name)
    cs, layer = _module_from_raw(raw, torch.float64)
    consts = layer.packed_constants()
    v = _to_my_basis(cs, csd, z["x"])
    y, kappa, _ = evaluate(consts, v)
    assert np.max(rel_err_rows(y, z["y64"])) < 1e-9
    # kappa of the normalised direction (what computeKappa returns in the reference)
    norm = np.linalg.norm(v, axis=1)
    ok = norm > 0
    kb = kappa[ok] / norm[ok]
    ref = z["kappa_bar64"][ok]
    assert np.max(np.abs(kb - ref) / np.maximum(1.0, np.abs(ref))) < 1e-8


@pytest.mark.parametrize("name", golden_names())
def test_packed_form_fp32_buffers_within_parity_bar(name):
    """Constants folded from fp32 buffers still reproduce the reference's fp32 output to 1e-5."""
    raw, csd, z = load_golden(name)
    cs, layer = _module_from_raw(raw, torch.float32)
    v = _to_my_basis(cs, csd, z["x"])
    y, _, _ = evaluate(layer.packed_constants(), v)
    assert np.max(rel_err_rows(y, z["y32"])) < 1e-5


def test_row_reductions():
    """Zero D rows dropped, tall SOC blocks QR-reduced, low-rank quadratics factored."""
    raw = workloads.corridor_like(k=20, n_eq=5, m=30, n_quad=3, rank=2, seed=1)
    cs, layer = _module_from_raw(raw, torch.float64)
    consts = layer.packed_constants()
    kinds = [s.type for s in consts.segments]
    assert kinds.count(_lib.SEG_QUAD_FAC) == 3
    for s in consts.segments:
        if s.type == _lib.SEG_QUAD_FAC:
            assert s.nrows == 3            # rank(P) + 1
    raw = workloads.random_lin_quad_soc(k=6, m=0, n_quad=0, n_soc=1, r_M=15, seed=3)
    cs, layer = _module_from_raw(raw, torch.float64)
    consts = layer.packed_constants()
    assert [s.type for s in consts.segments] == [_lib.SEG_SOC]       # the 0 z <= 1 filler row is gone
    assert consts.segments[0].nrows == 6                             # 15 x 6 block -> 6 x 6 factor
    assert consts.out_identity


def test_unsupported_methods_fail_loudly_and_host_tensors_stay_on_the_host():
    cs = workloads.build_constraints(workloads.cube())
    with pytest.raises(NotImplementedError):
        ConstraintModule(cs, method="DC3", create_map=False)
    assert ConstraintModule(cs, method="RAYEN_old", create_map=False).getDimAfterMap() == cs.n + 1
    with pytest.raises(RuntimeError):
        ConstraintModule(cs, create_map=True)              # input_dim missing (utils.verify)
    layer = ConstraintModule(cs, create_map=False)
    y = layer(torch.zeros(4, 3, 1))                        # host tensor: the packed torch evaluator (rayen_amd/eager.py)
    assert y.device.type == "cpu" and torch.equal(y[:, :, 0], torch.tensor(cs.y0.T, dtype=y.dtype).expand(4, 3))
    from rayen_amd import ops, pack as _pack
    with pytest.raises(RuntimeError, match="MI355X"):      # ... but the HIP ops themselves never take one
        ops._check_input(torch.zeros(4, 3), type("P", (), {"consts": layer.packed_constants(), "device_index": 0})())


def test_state_dict_and_pickle_roundtrip():
    import io
    cs = workloads.build_constraints(workloads.make_raw("c2"))
    layer = ConstraintModule(cs, input_dim=8, create_map=True)
    other = ConstraintModule(cs, input_dim=8, create_map=True)
    other.load_state_dict(layer.state_dict())
    assert torch.equal(other.mapper.weight, layer.mapper.weight)
    blob = io.BytesIO()
    torch.save(layer, blob)
    blob.seek(0)
    again = torch.load(blob, weights_only=False)
    assert torch.equal(again.all_delta, layer.all_delta)
    assert again.getDimAfterMap() == layer.getDimAfterMap()


def test_reference_import_names_resolve_to_this_build():
    """`from rayen import constraints, constraint_module` (the reference's import line) works here."""
    import rayen
    from rayen import constraint_module as cm, constraints as cons
    from rayen.constraint_module import ConstraintModule as CM2
    import rayen_amd
    assert cm.ConstraintModule is rayen_amd.constraint_module.ConstraintModule is CM2
    assert cons.ConvexConstraints is rayen_amd.constraints.ConvexConstraints
    assert rayen.utils.verify is rayen_amd.utils.verify


def test_low_rank_factoring_never_drops_curvature():
    """A quadratic is stored through its factor only when the fp64 constraint data is EXACTLY low rank.
    P = diag(1, 1e-6, 1e-6, 1e-6): the small eigenvalues are below fp32 resolution relative to the largest but
    they are curvature -- along e2 the reference's formula gives kappa = 5 at |v| = 5000 (clipping the output to
    the boundary); a factor that dropped them would return kappa = 0 and an infeasible y = v."""
    from rayen_amd import constraints
    P = np.diag([1.0, 1e-6, 1e-6, 1e-6])
    qc = constraints.ConvexQuadraticConstraint(P, np.zeros((4, 1)), np.array([[-0.5]]))
    cs = constraints.ConvexConstraints(qcs=[qc], y0=np.zeros((4, 1)))
    for dtype in (torch.float32, torch.float64):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            layer = ConstraintModule(cs, create_map=False)
        finally:
            torch.set_default_dtype(prev)
        consts = layer.packed_constants()
        assert [s.type for s in consts.segments] == [_lib.SEG_QUAD_SYM]
        v = np.zeros((1, 4))
        v[0, 1] = 5000.0
        y, kappa, _ = evaluate(consts, v)
        assert abs(kappa[0] - 5.0) < 1e-4
        assert 0.5 * y[0] @ P @ y[0] - 0.5 <= 1e-6   # (fp32 rounding of the 1e-6 entries)
    # the fp32 buffers of an exactly rank-3 form carry full-rank rounding noise: still factored, rank + 1 rows
    raw = workloads.corridor_like(k=20, n_eq=5, m=30, n_quad=3, rank=2, seed=1)
    cs32, layer32 = _module_from_raw(raw, torch.float32)
    fac = [s for s in layer32.packed_constants().segments if s.type == _lib.SEG_QUAD_FAC]
    assert len(fac) == 3 and all(s.nrows == 3 for s in fac)
    # buffers that do not come from the module's own constraint data (a foreign state_dict) are not trusted
    layer32.all_P.mul_(1.0 + 2.0 ** -10)
    layer32._invalidate_packs()
    kinds = [s.type for s in layer32.packed_constants().segments]
    assert _lib.SEG_QUAD_FAC not in kinds and kinds.count(_lib.SEG_QUAD_SYM) == 3


def test_config5_drops_the_same_segments_at_both_precisions():
    """the quadratics that vanish identically in the subspace are left out of the pack (kappa = 0); the
    decision is made on the fp64 constraint data against the module's own NA_E, so an fp32 module (rounded NA_E) and an
    fp64 one must agree -- 8 of config 5's 72 -- and the count is on record in the packed constants."""
    raw = workloads.make_raw("c5", seed=0)
    _, l32 = _module_from_raw(raw, torch.float32)
    _, l64 = _module_from_raw(raw, torch.float64)
    c32, c64 = l32.packed_constants(), l64.packed_constants()
    assert c32.dropped_segments == c64.dropped_segments == 8
    assert [(s.type, s.row0, s.nrows) for s in c32.segments] == [(s.type, s.row0, s.nrows) for s in c64.segments]
