"""The module's mapper fused into the projection kernel (``rayen_amd::ray_project_mapped``):
``ConstraintModule(cs, input_dim=..., create_map=True)`` = rayen/constraint_module.py:259-263 + :525 + :468-474
in one launch.  Checked against the CPU oracle fed with ``v = x W' + b`` and against the two-op path."""
import numpy as np
import pytest
import torch

from helpers import csd_from_cs, load_golden, rel_err_rows
from oracle import rayen_oracle as oracle
from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _sets():
    mixed_raw, _, _ = load_golden("example_13")            # lin + quad + SOC + LMI -> generic path, not fusable
    return {
        "c1": workloads.make_raw("c1"),                      # 6 rows in R^3: generic path (tiles too empty)
        "c2": workloads.make_raw("c2", seed=41),             # n = 16
        "c3": workloads.make_raw("c3", seed=42),             # n = 64, NA_E = I
        "c5r": workloads.make_raw("c5r", seed=43),             # n = 30 < k = 45: output through NA_E tiles
        "wide": workloads.random_lin_quad_soc(k=96, m=160, n_quad=2, n_soc=1, seed=44),   # n = 96 (NT = 1)
        "eq60": workloads.corridor_like(k=60, n_eq=10, m=120, n_quad=12, rank=3, seed=45),  # 32 < n = 50 <= 64, NA_E != I
        "ex13": mixed_raw,
    }


def _module(raw, input_dim, bias=True, seed=0):
    cs = workloads.build_constraints(raw)
    torch.manual_seed(seed)
    layer = ConstraintModule(cs, input_dim=input_dim, create_map=True)
    if not bias:
        layer.mapper = torch.nn.Linear(input_dim, cs.n, bias=False)
    return cs, layer.cuda()


def _oracle_y(cs, layer, x):
    w = layer.mapper.weight.detach().cpu().double()
    b = layer.mapper.bias.detach().cpu().double() if layer.mapper.bias is not None else 0.0
    v = (x.double() @ w.T + b).float()
    buf = oracle.precompute(csd_from_cs(cs), torch.float32)
    return oracle.forward(buf, v.unsqueeze(2)).numpy()[:, :, 0], v


@pytest.fixture(autouse=True, params=["default", "exact"])
def family(monkeypatch, request):
    """Every test of this file runs on both fp32 kernel families: the default one (split-operand kernel where the
    pack is eligible; its fused mapper reads a split-operand image of the weights, in_dim <= n rounded up to 32) and
    the exact-fp32 MFMA family (RAYEN_FP32_MODE=1, read when a pack is created; weights read in place, in_dim a
    multiple of 4 up to 64)."""
    if request.param == "exact":
        monkeypatch.setenv("RAYEN_FP32_MODE", "1")
    return request.param


@pytest.mark.parametrize("name,input_dim,fusable_exact,fusable_default", [
    ("c2", 8, True, True), ("c2", 64, True, False),          # n = 16: the default family keeps in_dim <= 32
    ("c3", 64, True, True), ("c3", 20, True, True), ("c3", 36, True, True),
    ("c5r", 32, True, True), ("c5r", 64, True, False),         # equality constraints (NA_E != I): the staged instances; in_dim <= 32 there
    ("c5r", 16, True, True), ("eq60", 40, True, True),        # n = 50 of k = 60 with 10 equalities: NKK = 2 staged, NKX = 2
    ("wide", 48, True, True),                                 # n = 96: exact-fp32 family in both runs
    ("c3", 6, False, True),                                   # not a multiple of 4: only the image form takes it
    ("c3", 96, False, False),                                 # wider than either fused kernel keeps in registers
    ("c1", 8, False, False), ("ex13", 8, False, False),       # packs on the generic path
])
def test_fused_mapper_matches_oracle_and_two_op_path(name, input_dim, fusable_exact, fusable_default, family):
    fusable = fusable_default if family == "default" else fusable_exact
    cs, layer = _module(_sets()[name], input_dim)
    B = 1000 if name != "c3" else 4133                        # ragged: not a multiple of 64
    gen = torch.Generator().manual_seed(5)
    x = torch.empty(B, input_dim).uniform_(-2.0, 2.0, generator=gen)
    x[:4] *= 1e-4
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    if family == "default" and name in ("c2", "c3", "c5r", "eq60"):
        assert dp.info().mfma_f32 == 3                        # the headline kernel, not a fallback
    assert ops.mapper_fusable(x.cuda(), layer.mapper.weight, layer.mapper.bias, dp) == fusable

    with torch.no_grad():
        y_fused = layer(x.cuda()).cpu().numpy()[:, :, 0]
        layer.fuse_mapper = False
        y_two = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref, _ = _oracle_y(cs, layer, x)
    assert y_fused.shape == (B, cs.k)
    # v itself differs by fp32 summation order between the MFMA chain, rocBLAS and the CPU: 1e-5 on y
    assert rel_err_rows(y_fused, y_ref).max() < 1e-5
    assert rel_err_rows(y_two, y_ref).max() < 1e-5
    viol = oracle.max_violation({**_sets()[name]}, y_fused.astype(np.float64))
    assert viol < max(1e-6, 3 * oracle.max_violation({**_sets()[name]}, y_ref.astype(np.float64)))


def test_weight_updates_reach_the_fused_kernel():
    """The default family reads the weights through an image that is rebuilt when they change (in-place update,
    optimiser step, load_state_dict): results must follow the weights immediately."""
    cs, layer = _module(_sets()["c3"], 64)
    x = torch.empty(500, 64).uniform_(-2.0, 2.0, generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        y_a = layer(x.cuda()).clone()
        layer.mapper.weight.mul_(0.5)
        layer.mapper.bias.add_(0.25)
        y_b = layer(x.cuda()).clone()
    assert not torch.equal(y_a, y_b)
    y_ref, _ = _oracle_y(cs, layer, x)
    assert rel_err_rows(y_b.cpu().numpy()[:, :, 0], y_ref).max() < 1e-5
    other = ConstraintModule(cs, input_dim=64, create_map=True).cuda()
    other.load_state_dict(layer.state_dict())
    with torch.no_grad():
        assert torch.equal(other(x.cuda()), y_b)
    # updates BEHIND the version counter (p.data.mul_, what some optimisers and EMA code do): parameters that are being
    # trained get a fresh image on every call ...
    layer.mapper.weight.data.mul_(1.5)
    with torch.no_grad():
        y_c = layer(x.cuda()).clone()
    y_ref, _ = _oracle_y(cs, layer, x)
    assert not torch.equal(y_c, y_b) and rel_err_rows(y_c.cpu().numpy()[:, :, 0], y_ref).max() < 1e-5
    # ... frozen ones are cached, and say so: invalidate_mapper_image() after such an update
    for prm in layer.mapper.parameters():
        prm.requires_grad_(False)
    with torch.no_grad():
        layer(x.cuda())                                   # (builds and caches the image of the frozen weights)
        layer.mapper.weight.data.mul_(0.5)
        dp, _ = layer.device_pack(torch.device("cuda", 0))
        dp.invalidate_mapper_image()
        y_d = layer(x.cuda()).clone()
    y_ref, _ = _oracle_y(cs, layer, x)
    assert rel_err_rows(y_d.cpu().numpy()[:, :, 0], y_ref).max() < 1e-5


def test_no_bias_and_strided_input():
    cs, layer = _module(_sets()["c3"], 32, bias=False)
    gen = torch.Generator().manual_seed(6)
    big = torch.empty(777, 40).uniform_(-1.5, 1.5, generator=gen)
    x = big[:, :32]                                            # row stride 40 floats: rows stay 16-byte aligned
    with torch.no_grad():
        y = layer(x.cuda()).cpu().numpy()[:, :, 0]
    y_ref, _ = _oracle_y(cs, layer, x)
    assert rel_err_rows(y, y_ref).max() < 1e-5


def test_v_is_written_only_for_training_and_equals_the_linear_map():
    cs, layer = _module(_sets()["c3"], 64)
    x = torch.randn(300, 64, device="cuda")
    _, pack_id = layer.device_pack(x.device)
    y, kappa, active, v = torch.ops.rayen_amd.ray_project_mapped(x, layer.mapper.weight, layer.mapper.bias, pack_id, True)
    want = torch.nn.functional.linear(x.double(), layer.mapper.weight.double(), layer.mapper.bias.double())
    assert torch.allclose(v.detach().double(), want.detach(), rtol=0, atol=2e-6 * float(want.detach().abs().max()))
    assert active.shape == (300, 2) and int(active[:, 0].max()) >= 0
    with torch.no_grad():
        y2, kappa2, active2, v2 = torch.ops.rayen_amd.ray_project_mapped(x, layer.mapper.weight, layer.mapper.bias, pack_id, False)
    assert v2.numel() == 0 and active2.numel() == 0
    assert torch.equal(y.detach(), y2) and torch.equal(kappa.detach(), kappa2)


@pytest.mark.parametrize("name,input_dim", [("c3", 64), ("c2", 8), ("c5r", 32), ("eq60", 40)])
def test_gradients_of_the_fused_layer_match_the_two_op_path(name, input_dim):
    cs, layer = _module(_sets()[name], input_dim)
    gen = torch.Generator().manual_seed(8)
    x = torch.empty(640, input_dim).uniform_(-2.0, 2.0, generator=gen)
    G = torch.empty(640, cs.k, 1).uniform_(-1.0, 1.0, generator=gen).cuda()

    def grads(fuse):
        layer.fuse_mapper = fuse
        layer.zero_grad()
        xg = x.cuda().requires_grad_(True)
        (layer(xg) * G).sum().backward()
        return xg.grad.cpu(), layer.mapper.weight.grad.cpu().clone(), layer.mapper.bias.grad.cpu().clone()

    gx1, gw1, gb1 = grads(True)
    gx0, gw0, gb0 = grads(False)
    # identical backward kernel on v that differs in the last bits: a sample on a kink of kappa may flip
    bad_rows = (gx1 - gx0).abs().amax(1) > 1e-4 * gx0.abs().amax().clamp_min(1e-12)
    assert bad_rows.float().mean() < 0.01
    # the weight / bias gradients are sums over the batch: the few kink rows above are all that may differ
    good = ~bad_rows
    assert torch.allclose(gx1[good], gx0[good], rtol=0, atol=1e-4 * float(gx0.abs().max()))
    assert torch.allclose(gw1, gw0, rtol=0, atol=2e-2 * float(gw0.abs().max()))
    assert torch.allclose(gb1, gb0, rtol=0, atol=2e-2 * float(gb0.abs().max()))


def test_training_with_the_fused_layer_follows_the_two_op_trajectory():
    cs = workloads.build_constraints(_sets()["c2"])

    def run(fuse):
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(),
                                    ConstraintModule(cs, input_dim=32, create_map=True)).cuda()
        model[2].fuse_mapper = fuse
        opt = torch.optim.SGD(model.parameters(), lr=5e-2)
        gen = torch.Generator().manual_seed(1)
        x = torch.randn(256, 6, generator=gen).cuda()
        target = (0.05 * torch.randn(256, cs.k, 1, generator=gen)).cuda()
        losses = []
        for _ in range(25):
            opt.zero_grad()
            loss = ((model(x) - target) ** 2).mean()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return np.array(losses)

    fused, two_op = run(True), run(False)
    assert np.all(np.isfinite(fused)) and fused[-1] < fused[0]
    assert np.allclose(fused, two_op, rtol=1e-3, atol=0)


def test_fused_layer_under_hip_graph_capture():
    cs, layer = _module(_sets()["c3"], 64)
    layer.check_nan = False
    x = torch.randn(2048, 64, device="cuda")
    with torch.no_grad():
        eager = layer(x).clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            layer(x)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = layer(x)
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
