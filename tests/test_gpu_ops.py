"""Custom-op hygiene on the GPU: torch.library.opcheck (schema, fake kernels, autograd registration),
bitwise reproducibility, concurrent use of one constant pack from several streams and threads."""
import threading

import pytest
import torch

from rayen_amd import ops, workloads
from rayen_amd.constraint_module import ConstraintModule

pytestmark = pytest.mark.gpu


def _layer(name, **kw):
    cs = workloads.build_constraints(workloads.make_raw(name, seed=3))
    return cs, ConstraintModule(cs, **kw).cuda()


@pytest.mark.parametrize("name", ["c2", "c4", "c5r"])
def test_opcheck_ray_project(name):
    cs, layer = _layer(name, create_map=False)
    _, pack_id = layer.device_pack(torch.device("cuda", 0))
    v = torch.empty(97, cs.n, device="cuda").uniform_(-1.5, 1.5).requires_grad_(True)
    torch.library.opcheck(torch.ops.rayen_amd.ray_project.default, (v, pack_id, True, False),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))
    y, kappa, active = torch.ops.rayen_amd.ray_project(v.detach(), pack_id, True, False)
    g = torch.randn_like(y)
    torch.library.opcheck(torch.ops.rayen_amd.ray_project_bwd.default, (v.detach(), kappa, active, g, pack_id, False),
                          test_utils=("test_schema", "test_faketensor"))


@pytest.mark.parametrize("family", ["default", "exact"])
def test_opcheck_ray_project_mapped(monkeypatch, family):
    if family == "exact":
        monkeypatch.setenv("RAYEN_FP32_MODE", "1")   # weights read in place by the exact-fp32 MFMA family
    cs, layer = _layer("c3", input_dim=32, create_map=True)
    _, pack_id = layer.device_pack(torch.device("cuda", 0))
    x = torch.randn(130, 32, device="cuda", requires_grad=True)
    torch.library.opcheck(torch.ops.rayen_amd.ray_project_mapped.default,
                          (x, layer.mapper.weight, layer.mapper.bias, pack_id, True),
                          test_utils=("test_schema", "test_faketensor", "test_autograd_registration"))


@pytest.mark.parametrize("name", ["c3", "c4", "c5r"])
def test_bitwise_reproducible(name):
    cs, layer = _layer(name, create_map=False)
    x = torch.empty(5000, cs.n, 1, device="cuda").uniform_(-1, 1)
    with torch.no_grad():
        a = layer(x).clone()
        for _ in range(3):
            assert torch.equal(layer(x), a)


def test_one_pack_many_streams_and_threads():
    cs, layer = _layer("c3", create_map=False)
    dp, _ = layer.device_pack(torch.device("cuda", 0))
    xs = [torch.empty(20000 + 777 * i, cs.n, device="cuda").uniform_(-1, 1) for i in range(6)]
    with torch.no_grad():
        want = [ops.project_raw(x, dp, want_active=False)[0].clone() for x in xs]
    torch.cuda.synchronize()
    got = [None] * len(xs)

    def work(i):
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream), torch.no_grad():
            for _ in range(5):
                got[i] = ops.project_raw(xs[i], dp, want_active=False)[0]
        stream.synchronize()

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for g, w in zip(got, want):
        assert torch.equal(g, w)


def test_pack_create_destroy_does_not_leak():
    cs = workloads.build_constraints(workloads.make_raw("c3", seed=9))
    import gc
    x = torch.empty(4096, cs.n, 1, device="cuda").uniform_(-1, 1)

    def cycle(times):
        for _ in range(times):
            layer = ConstraintModule(cs, create_map=False).cuda()
            xr = x.clone().requires_grad_(True)
            layer(xr).sum().backward()
            del layer, xr
        gc.collect()                                 # the module holds a bound method of itself: a cycle
        torch.cuda.synchronize()
        return torch.cuda.mem_get_info()[0]

    cycle(2)                                         # one-time allocations (allocator pools, code objects)
    free0 = cycle(1)
    free1 = cycle(40)
    assert free0 - free1 < 8 * 1024 * 1024           # every pack holds ~0.5 MB of device images
