"""The drop-in boundary on the GPU: a pack is complete when ``rayen_pack_create`` returns (no first-call
allocation, no lock: capture-safe and thread-safe from its first call), precision masks fail loudly, and the
solver-free set-up (no ``y0``) feeds the same HIP forward."""
import ctypes
import threading

import numpy as np
import pytest
import torch

from helpers import csd_from_cs, load_golden, rel_err_rows
from oracle import rayen_oracle as oracle
from rayen_amd import _lib, constraints, ops, pack as _pack, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _cold_layer(name, dtype=torch.float32, seed=3):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cs = workloads.build_constraints(workloads.make_raw(name, seed=seed))
        return cs, ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)


@pytest.mark.parametrize("name", ["c2", "c3", "c4", "c5r"])
def test_cold_pack_is_captured_into_a_hip_graph(name):
    """The very FIRST projection call of a pack happens inside a stream capture (forward with the arg-max record,
    and the backward): nothing may allocate, synchronise or touch another stream.  Replays must equal a plain call."""
    cs, layer = _cold_layer(name)
    dp, _ = layer.device_pack(torch.device("cuda", 0))          # rayen_pack_create runs here, outside the capture
    mem_before = dp.info().device_bytes
    gen = torch.Generator(device="cuda").manual_seed(5)
    v = torch.empty(3000, cs.n, device="cuda").uniform_(-1.5, 1.5, generator=gen)
    g = torch.empty(3000, cs.k, device="cuda").uniform_(-1, 1, generator=gen)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        # buffers of the captured region are allocated by torch's graph pool; the C ABI itself allocates nothing
        with torch.cuda.graph(graph, stream=side):
            y, kappa, active = ops.project_raw(v, dp, want_active=True)
            y_plain, _, _ = ops.project_raw(v, dp, want_active=False)
            grad = ops.backward_raw(v, kappa, active, g, dp)
    torch.cuda.current_stream().wait_stream(side)
    for trial in range(2):
        v.uniform_(-1.5, 1.5, generator=gen)
        graph.replay()
        torch.cuda.synchronize()
        y_ref, kappa_ref, active_ref = ops.project_raw(v, dp, want_active=True)
        assert torch.equal(y, y_ref) and torch.equal(kappa, kappa_ref) and torch.equal(active, active_ref)
        assert torch.equal(y_plain, ops.project_raw(v, dp, want_active=False)[0])
        assert torch.equal(grad, ops.backward_raw(v, kappa_ref, active_ref, g, dp))
    assert dp.info().device_bytes == mem_before                 # no image appeared after creation
    x = v[:512].cpu().unsqueeze(2)
    y_true = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float32), x).numpy()[:, :, 0]
    assert np.max(rel_err_rows(y[:512].cpu().numpy(), y_true)) <= 1e-5


@pytest.mark.parametrize("name,dtype", [("c3", torch.float32), ("c5r", torch.float32), ("c4", torch.float32),
                                        ("c3", torch.float64)])
def test_cold_pack_hammered_by_eight_threads(name, dtype):
    """Eight threads, each on its own stream, make their first calls on a pack nobody has used yet (forward, tracked
    forward, backward): results equal the single-threaded ones bit for bit."""
    cs, layer = _cold_layer(name, dtype)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    gen = torch.Generator(device="cuda").manual_seed(8)
    vs = [torch.empty(9000 + 331 * i, cs.n, device="cuda", dtype=dtype).uniform_(-1.5, 1.5, generator=gen) for i in range(8)]
    gs = [torch.empty(v.shape[0], cs.k, device="cuda", dtype=dtype).uniform_(-1, 1, generator=gen) for v in vs]
    torch.cuda.synchronize()
    out, errors = [None] * 8, []
    start = threading.Barrier(8)

    def work(i):
        try:
            stream = torch.cuda.Stream()
            start.wait()
            with torch.cuda.stream(stream), torch.no_grad():
                for _ in range(5):
                    y0_, _, _ = ops.project_raw(vs[i], dp, want_active=False)
                    y, kappa, active = ops.project_raw(vs[i], dp, want_active=True)
                    grad = ops.backward_raw(vs[i], kappa, active, gs[i], dp)
            stream.synchronize()
            out[i] = (y0_, y, kappa, active, grad)
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i in range(8):
        y0_, _, _ = ops.project_raw(vs[i], dp, want_active=False)
        y, kappa, active = ops.project_raw(vs[i], dp, want_active=True)
        grad = ops.backward_raw(vs[i], kappa, active, gs[i], dp)
        for a, b in zip(out[i], (y0_, y, kappa, active, grad)):
            assert torch.equal(a, b), (name, i)


def test_prepare_mask_and_fp32_mode():
    """RayenPackDesc.prepare limits what pack_create builds; calls into a family left out fail loudly.
    RayenPackDesc.fp32_mode pins the fp32 forward family without the environment variable."""
    cs, layer = _cold_layer("c3")
    consts = layer.packed_constants()
    v32 = torch.empty(100, cs.n, device="cuda").uniform_(-1, 1)
    v64 = v32.double()
    only32 = _pack.DevicePack(consts, 0, prepare=_lib.PREPARE_F32 | _lib.PREPARE_FWD_ONLY)
    info = only32.info()
    assert info.prepared == _lib.PREPARE_F32
    y, kappa, active = ops.project_raw(v32, only32, want_active=True)
    with pytest.raises(_lib.RayenError) as exc:
        ops.project_raw(v64, only32)
    assert exc.value.code == -8
    with pytest.raises(_lib.RayenError) as exc:
        ops.backward_raw(v32, kappa, active, torch.ones(100, cs.k, device="cuda"), only32)
    assert exc.value.code == -8
    # RAYEN_PREPARE_FWD_ONLY alone: no precision bit = both precisions, forward only
    fwd_only = _pack.DevicePack(consts, 0, prepare=_lib.PREPARE_FWD_ONLY)
    assert fwd_only.info().prepared == (_lib.PREPARE_F32 | _lib.PREPARE_F64)
    assert torch.equal(ops.project_raw(v32, fwd_only)[0], y)
    ops.project_raw(v64, fwd_only)
    with pytest.raises(_lib.RayenError) as exc:
        ops.backward_raw(v32, kappa, active, torch.ones(100, cs.k, device="cuda"), fwd_only)
    assert exc.value.code == -8
    fwd_only.close()
    full = _pack.DevicePack(consts, 0)
    assert full.info().prepared == 7 and full.info().device_bytes > info.device_bytes
    assert torch.equal(ops.project_raw(v32, full)[0], y)
    exact = _pack.DevicePack(consts, 0, fp32_mode=1)
    triple_unchecked = _pack.DevicePack(consts, 0, fp32_mode=2)
    pair_unchecked = _pack.DevicePack(consts, 0, fp32_mode=3)
    no_pair = _pack.DevicePack(consts, 0, fp32_mode=4)
    packs = (full, exact, triple_unchecked, pair_unchecked, no_pair)
    assert tuple(dp.info().mfma_f32 for dp in packs) == (3, 1, 2, 3, 2)
    assert full.info().fp32_check_pair >= 0 and full.info().fp32_check_split >= 0 and full.info().fp32_check_exact >= 0
    assert exact.info().fp32_check_split == -1.0 and pair_unchecked.info().fp32_check_pair == -1.0
    assert no_pair.info().fp32_check_pair == -1.0 and no_pair.info().fp32_check_split >= 0
    y_true = ops.project_raw(v64, full)[0]
    for dp in packs:
        assert np.max(rel_err_rows(ops.project_raw(v32, dp)[0].cpu().numpy(), y_true.cpu().numpy())) <= 1e-5
    for dp in (only32,) + packs:
        dp.close()


# ---------------------------------------------------------------------------------------------------
# sets built WITHOUT y0: solver-free preprocessing (scipy LPs + rayen_amd/conic.py) feeding the HIP forward
# ---------------------------------------------------------------------------------------------------

def _build_without_y0(raw):
    lc = None
    if raw["A1"] is not None or raw["A2"] is not None:
        lc = constraints.LinearConstraint(raw["A1"], raw["b1"], raw["A2"], raw["b2"])
    qcs = [constraints.ConvexQuadraticConstraint(P, q, r) for P, q, r in zip(raw["P"], raw["q"], raw["r"])]
    socs = [constraints.SOCConstraint(M, s, c, d) for M, s, c, d in zip(raw["M"], raw["s"], raw["c"], raw["d"])]
    lmic = constraints.LMIConstraint(list(raw["F"])) if len(raw["F"]) else None
    return constraints.ConvexConstraints(lc=lc, qcs=qcs, socs=socs, lmic=lmic)


@pytest.mark.parametrize("index", range(15))
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.float64, 1e-9)])
def test_example_sets_without_y0_on_the_hip_forward(index, dtype, tol):
    """examples/examples_sets.py:94-194 the way examples/test_layer.py builds them (no y0, linear preprocessing on):
    the interior point is this build's own (parity unpinned for z0), so the oracle is fed the SAME cs -- the forward
    must match it, every output must be feasible, and getViolation (constraints.py:549-559) must be ~0 on outputs."""
    raw, _, _ = load_golden(f"example_{index:02d}")
    cs = _build_without_y0(raw)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        layer = ConstraintModule(cs, create_map=False).cuda()
    finally:
        torch.set_default_dtype(prev)
    gen = torch.Generator().manual_seed(index)
    x = torch.empty(500, cs.n, 1, dtype=torch.float32).uniform_(-5, 5, generator=gen).to(dtype)   # test_layer.py:74-75
    x[:2] *= 1e-4
    y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref = oracle.forward(oracle.precompute(csd_from_cs(cs), dtype), x).numpy()[:, :, 0]
    if dtype == torch.float32:
        y_true = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x.double()).numpy()[:, :, 0]
        bound = max(tol, 2.0 * rel_err_rows(y_ref, y_true).max())
        assert rel_err_rows(y, y_true).max() <= bound
    else:
        assert np.max(rel_err_rows(y, y_ref)) <= tol
    floor = 1e-6 if dtype == torch.float32 else 1e-11
    assert cs.getMaxViolation(y) <= max(floor, 3 * cs.getMaxViolation(y_ref))
    # the reference's own violation measure on a few outputs: squared distance to the set
    for row in y[:3]:
        assert cs.getViolation(row.astype(np.float64)) <= (1e-10 if dtype == torch.float32 else 1e-14)
    far = 50.0 * np.ones(cs.k)
    assert cs.getViolation(far) > 1e-3 or cs.getMaxViolation(far[None]) <= 0


def test_reserved_compute_units_change_the_launch_not_the_values():
    """``rayen_reserve_cus`` (ABI v4) shrinks the persistent grids of the matrix-core kernels so that a collective can
    run beside them (the multi-GPU gather step); outputs are the same bits, and the setting is restored by its caller."""
    lib = _lib.load()
    assert lib.rayen_reserve_cus(-1) == 0
    for name in ("c3", "c5r"):
        cs, layer = _cold_layer(name)
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        v = torch.empty(300000, cs.n, device="cuda").uniform_(-1.5, 1.5)
        y0, k0, a0 = ops.project_raw(v, dp, want_active=True)
        g = torch.randn(v.shape[0], cs.k, device="cuda")
        grad0 = ops.backward_raw(v, k0, a0, g, dp)
        for cus in (8, 64, 255):
            prev = lib.rayen_reserve_cus(cus)
            try:
                y1, k1, a1 = ops.project_raw(v, dp, want_active=True)
                grad1 = ops.backward_raw(v, k1, a1, g, dp)
            finally:
                lib.rayen_reserve_cus(prev)
            assert torch.equal(y0, y1) and torch.equal(k0, k1) and torch.equal(a0, a1) and torch.equal(grad0, grad1)
    assert lib.rayen_reserve_cus(-1) == 0


@pytest.mark.eager_detour
def test_a_set_no_kernel_serves_is_evaluated_by_the_packed_torch_evaluator_loudly(monkeypatch):
    """an LMI above 30 x 30 mixed with a quadratic has no HIP kernel.  The module says so once
    (RuntimeWarning) and evaluates the packed form with torch ops ON THE DEVICE (rayen_amd/eager.py);
    RAYEN_STRICT_HIP=1 keeps the C ABI's error.  A set the kernels serve never takes this path (every other test)."""
    import warnings
    from helpers import csd_from_cs
    from oracle import rayen_oracle as oracle
    from rayen_amd import _lib
    # (fp64, an LMI of 230 x 230 next to a quadratic: the workgroup-per-sample kernel holds r <= 212 in fp64 -- the 40 x 40
    # of rounds 3-4 is served since round 5, tests/test_gpu_lmi_mixed.py)
    raw = workloads.random_lmi(5, 230, seed=2)
    rng = np.random.default_rng(0)
    T = rng.uniform(-1, 1, size=(5, 5))
    raw["P"], raw["q"], raw["r"] = [T @ T.T], [rng.uniform(-1, 1, size=(5, 1))], [np.array([[-0.5]])]
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        cs = workloads.build_constraints(raw)
        layer = ConstraintModule(cs, create_map=False).cuda()
        x = torch.empty(64, cs.n, 1, dtype=torch.float64).uniform_(-2, 2)
        monkeypatch.setenv("RAYEN_STRICT_HIP", "1")
        with pytest.raises(_lib.RayenError):
            layer(x.cuda())
        monkeypatch.delenv("RAYEN_STRICT_HIP")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            y = layer(x.cuda())
            layer(x.cuda())
        assert sum(issubclass(w.category, RuntimeWarning) for w in caught) == 1
        assert y.is_cuda and layer._hip_unsupported
        want = oracle.forward(oracle.precompute(csd_from_cs(cs), torch.float64), x)
        assert float((y.cpu() - want).abs().max()) < 1e-9
        xg = x.cuda().requires_grad_(True)
        layer(xg).sum().backward()
        assert torch.isfinite(xg.grad).all()
    finally:
        torch.set_default_dtype(prev)
