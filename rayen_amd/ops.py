"""``torch.library`` custom ops in front of the C ABI (``include/rayen_hip.h``).

``rayen_amd::ray_project(v, pack_id) -> (y, kappa, active)`` is the fused
replacement of ``forwardForRAYEN`` (rayen/constraint_module.py:468-474); it is
asynchronous on torch's current HIP stream and differentiable (the backward is
``rayen_amd::ray_project_bwd``, another HIP kernel).  PyTorch is used here for
device memory and the stream only.

These ops serve tensors on a HIP device only (anything else raises HERE; ``ConstraintModule`` sends host tensors
to ``rayen_amd/eager.py`` before it gets this far).  One detour exists and it is loud: a backward the kernels
decline with ``RAYEN_E_UNSUPPORTED`` (n beyond what they stage) is evaluated by autograd through the
packed torch evaluator on the same device, with a ``RuntimeWarning``; ``RAYEN_STRICT_HIP=1`` keeps the error.
"""
from __future__ import annotations

import ctypes
import os
import warnings
import weakref
from typing import Optional

import torch

from . import _lib

_packs = weakref.WeakValueDictionary()
_next_id = [1]


def register_pack(pack) -> int:
    pack_id = _next_id[0]
    _next_id[0] += 1
    _packs[pack_id] = pack
    return pack_id


def _pack(pack_id):
    pack = _packs.get(pack_id)
    if pack is None or pack.handle is None:
        raise RuntimeError(f"rayen_amd: constant pack {pack_id} no longer exists")
    return pack


def _check_input(v, pack):
    if not v.is_cuda:
        raise RuntimeError("rayen_amd: the projection runs on an MI355X (HIP) device only; got a "
                           f"{v.device} tensor. Move the module and its input to 'cuda'.")
    if v.dtype not in (torch.float32, torch.float64):
        raise RuntimeError(f"rayen_amd: unsupported dtype {v.dtype} (float32 and float64 only)")
    if v.dim() != 2 or v.shape[1] < pack.consts.n:
        raise RuntimeError(f"rayen_amd: expected v of shape [B, >= {pack.consts.n}], got {tuple(v.shape)}")
    if v.device.index != pack.device_index:
        raise RuntimeError("rayen_amd: input and constant pack live on different devices")


def _ptr(t):
    """Device address of a tensor as a plain int (ctypes converts it to ``void*``; building a ``c_void_p`` object per
    argument costs ~0.3 us each -- ten of them per call is a third of a small-batch forward's host time)."""
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(index=None):
    """The caller's current HIP stream (a raw hipStream_t).  ``torch.cuda.current_stream().cuda_stream`` costs
    several microseconds of Python per call, which is most of a small-batch forward; the private accessor is what
    it ends in."""
    if _raw_stream is not None and index is not None:
        return _raw_stream(index) or None
    return torch.cuda.current_stream().cuda_stream or None


class _on_device:
    """``with torch.cuda.device(d)`` only when ``d`` is not already the current device."""
    __slots__ = ("index", "ctx")

    def __init__(self, device):
        self.index = device.index
        self.ctx = None

    def __enter__(self):
        if torch.cuda.current_device() != self.index:
            self.ctx = torch.cuda.device(self.index)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


_ENTRY = {}


def _entry(name):
    """The ctypes function object of an entry point (looked up once)."""
    fn = _ENTRY.get(name)
    if fn is None:
        fn = _ENTRY[name] = getattr(_lib.load(), name)
    return fn


_FWD = {(torch.float32, False): "rayen_ray_project_f32", (torch.float64, False): "rayen_ray_project_f64",
        (torch.float32, True): "rayen_ray_project_generic_f32",
        (torch.float64, True): "rayen_ray_project_generic_f64"}
_FWD_OLD = {torch.float32: "rayen_ray_project_old_f32", torch.float64: "rayen_ray_project_old_f64"}
_BWD = {(torch.float32, False): "rayen_ray_project_bwd_f32", (torch.float64, False): "rayen_ray_project_bwd_f64",
        (torch.float32, True): "rayen_ray_project_old_bwd_f32",
        (torch.float64, True): "rayen_ray_project_old_bwd_f64"}


# Where the matrix-core kernels stop (n they keep in registers; include/rayen_hip.h) the forward
# goes GEMM + epilogue: T = v W_ext' on the vendor library (torch.mm -> hipBLASLt / rocBLAS on the caller's stream),
# then ONE hand-written kernel over T (rayen_wide.hip).  ``RAYEN_WIDE_ROUTE=0`` pins the lane-per-sample kernel.
_WIDE_MIN_N = {torch.float32: 129, torch.float64: 65}
# Sets = [linear rows] + one LMI on the workgroup-per-sample kernels (round 5) take the same route much earlier: S(v) for the
# whole batch is then ONE GEMM instead of every workgroup streaming all k generators for its sample.  Measured, B = 2 000, fp32,
# forward / backward ms, products against fused (profiles/bench/r05_lmi_products_ab.txt): r = 100 k = 50 0.84 / 0.99 against
# 0.89 / 1.14, k = 100 0.85 / 1.05 against 0.95 / 1.32, k = 1 000 1.01 / 1.31 against 2.34 / 7.72; r = 300 k = 100 14.9 / 17.3
# against 19.2 / 25.9, k = 500 15.6 / 18.5 against 37.8 / 63.2; a tie at k = 10.  T and the backward's C are [B, rows of W_ext]
# each: beyond _LMI_PRODUCTS_BYTES per matrix the fused kernels keep the batch (they need no scratch).
_LMI_WIDE_MIN_N = 32
_LMI_PRODUCTS_BYTES = 4 << 30


def _wide_route(v, pack, force_generic, old_head):
    if force_generic or old_head or os.environ.get("RAYEN_WIDE_ROUTE", "1") == "0":
        return None
    env = os.environ.get("RAYEN_WIDE_MIN_N")          # (developer A/B)
    has_lmi = any(seg.type == _lib.SEG_LMI for seg in pack.consts.segments)
    min_n = int(env) if env else (_LMI_WIDE_MIN_N if has_lmi else _WIDE_MIN_N[v.dtype])
    if pack.consts.n < min_n:
        return None
    Wt = pack.products_matrix(v.dtype)
    if Wt is not None and has_lmi and v.shape[0] * Wt.shape[1] * v.element_size() > _LMI_PRODUCTS_BYTES:
        return None
    return Wt


def project_raw(v, pack, want_y=True, force_generic=False, want_active=True, old_head=False, out=None,
                want_kappa=True):
    """Direct call of the C ABI on an existing ``DevicePack``; returns (y|None, kappa|None, active|None).

    ``old_head``: the ``RAYEN_old`` step rule; ``v`` then carries ``beta`` in column ``n``.
    ``out``: a ``[B, >=k]`` tensor (unit column stride) whose first ``k`` columns receive ``y`` -- e.g. this
    rank's rows of a gather buffer -- instead of a fresh allocation."""
    _check_input(v, pack)
    if old_head and v.shape[1] < pack.consts.n + 1:
        raise RuntimeError(f"rayen_amd: RAYEN_old needs {pack.consts.n + 1} input columns, got {v.shape[1]}")
    if v.stride(1) != 1:
        v = v.contiguous()
    B = v.shape[0]
    k = pack.consts.k
    if out is not None:
        if (out.dim() != 2 or out.shape[0] != B or out.shape[1] < k or out.dtype != v.dtype
                or out.device != v.device or (B and out.stride(1) != 1)):
            raise RuntimeError(f"rayen_amd: out must be a [{B}, >={k}] {v.dtype} tensor on {v.device} with unit column stride")
        y = out
    else:
        y = torch.empty((B, k), dtype=v.dtype, device=v.device) if want_y else None
    kappa = torch.empty((B,), dtype=v.dtype, device=v.device) if want_kappa else None
    active = torch.empty((B, 2), dtype=torch.int32, device=v.device) if want_active else None
    Wt = _wide_route(v, pack, force_generic, old_head) if B else None
    with _on_device(v.device):
        if Wt is not None:
            n = pack.consts.n
            prods = torch.mm(v if v.shape[1] == n else v[:, :n], Wt)         # [B, rows of W_ext]: the library GEMM
            fn = _entry("rayen_ray_project_from_products_f32" if v.dtype == torch.float32
                        else "rayen_ray_project_from_products_f64")
            code = fn(pack.handle, _ptr(prods), prods.stride(0), _ptr(v), B, v.stride(0), _ptr(y),
                      y.stride(0) if y is not None else k, _ptr(kappa), _ptr(active), _ptr(pack.nan_flag),
                      _stream(v.device.index))
        else:
            fn = _entry(_FWD_OLD[v.dtype] if old_head else _FWD[(v.dtype, bool(force_generic))])
            code = fn(pack.handle, _ptr(v), B, v.stride(0) if B else pack.consts.n, _ptr(y),
                      y.stride(0) if (y is not None and B) else k,
                      _ptr(kappa), _ptr(active), _ptr(pack.nan_flag), _stream(v.device.index))
    _lib.check(code, "rayen_ray_project")
    return y, kappa, active


@torch.library.custom_op("rayen_amd::ray_project", mutates_args=())
def ray_project(v: torch.Tensor, pack_id: int, need_active: bool, old_head: bool = False) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """``need_active``: also record which constraint set kappa (only the backward reads it; without
    it the kernels skip the arg-max bookkeeping and ``active`` comes back empty).
    ``old_head``: the ``RAYEN_old`` step rule (``v[:, n]`` is ``beta``)."""
    y, kappa, active = project_raw(v, _pack(pack_id), want_active=need_active, old_head=old_head)
    if active is None:
        active = torch.empty((0, 2), dtype=torch.int32, device=v.device)
    return y, kappa, active


@ray_project.register_fake
def _(v, pack_id, need_active, old_head=False):
    pack = _pack(pack_id)
    B = v.shape[0]
    return (v.new_empty((B, pack.consts.k)), v.new_empty((B,)),
            v.new_empty((B if need_active else 0, 2), dtype=torch.int32))


def backward_raw(v, kappa, active, grad_y, pack, old_head=False, force_generic=False, bucketed=True):
    """Direct call of the backward entry points; ``force_generic`` pins the lane-per-sample kernel
    where ``rayen_ray_project_bwd_f32/_f64`` would pick the matrix-core one.  ``bucketed``: hand the library the
    scratch buffer it asks for (``rayen_bwd_workspace_bytes_f32``) so that it may group the samples by active
    constraint and walk only that constraint's tiles; ``False`` pins the plain walk (same results)."""
    _check_input(v, pack)
    if v.stride(1) != 1:
        v = v.contiguous()
    grad_y = grad_y.contiguous()
    B = v.shape[0]
    # the kernels write the first n (+1 for the old head) columns of every row; wider inputs keep zeros
    used = pack.consts.n + (1 if old_head else 0)
    grad_v = torch.empty_like(v) if v.shape[1] == used else torch.zeros_like(v)
    name = _BWD[(v.dtype, bool(old_head))]
    if force_generic:
        if old_head:
            raise RuntimeError("force_generic selects between the two RAYEN backward kernels only")
        name = "rayen_ray_project_bwd_generic_f32" if v.dtype == torch.float32 else "rayen_ray_project_bwd_generic_f64"
    Wt = _wide_route(v, pack, force_generic, old_head) if B else None
    if Wt is not None:
        # wide sets: T = v W_ext' again (one GEMM), the coefficient kernel, grad_v = C W_ext (+ s g): rayen_wide.hip
        n, k = pack.consts.n, pack.consts.k
        identity = bool(pack.consts.out_identity)
        with _on_device(v.device):
            prods = torch.mm(v if v.shape[1] == n else v[:, :n], Wt)
            coeff = torch.empty_like(prods)
            gs = torch.empty((B, k), dtype=v.dtype, device=v.device) if identity else None
            fn = _entry("rayen_ray_project_bwd_coefficients_f32" if v.dtype == torch.float32
                        else "rayen_ray_project_bwd_coefficients_f64")
            code = fn(pack.handle, _ptr(prods), prods.stride(0), _ptr(v), B, v.stride(0), _ptr(kappa), _ptr(active),
                      _ptr(grad_y), grad_y.stride(0), _ptr(coeff), coeff.stride(0), _ptr(gs), _stream(v.device.index))
            _lib.check(code, "rayen_ray_project_bwd_coefficients")
            core = torch.addmm(gs, coeff, Wt.t()) if identity else torch.mm(coeff, Wt.t())
        if v.shape[1] == n:
            return core
        grad_v[:, :n] = core
        return grad_v
    with _on_device(v.device):
        lib = _lib.load()
        ws_bytes = 0
        tag = "f32" if v.dtype == torch.float32 else "f64"
        if bucketed and not old_head and not force_generic and B:
            ws_bytes = int(getattr(lib, "rayen_bwd_workspace_bytes_" + tag)(pack.handle, B))
        if ws_bytes > 0:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=v.device)   # (torch's caching allocator: no hipMalloc)
            code = getattr(lib, "rayen_ray_project_bwd_ws_" + tag)(pack.handle, _ptr(v), B, v.stride(0), _ptr(kappa), _ptr(active),
                                                                   _ptr(grad_y), grad_y.shape[1], _ptr(grad_v), grad_v.stride(0),
                                                                   _ptr(ws), ws_bytes, _stream(v.device.index))
        else:
            code = getattr(lib, name)(pack.handle, _ptr(v), B, v.stride(0) if B else pack.consts.n,
                                      _ptr(kappa), _ptr(active), _ptr(grad_y), grad_y.shape[1],
                                      _ptr(grad_v), grad_v.stride(0) if B else pack.consts.n,
                                      _stream(v.device.index))
    _lib.check(code, "rayen_ray_project_bwd")
    return grad_v


@torch.library.custom_op("rayen_amd::ray_project_bwd", mutates_args=())
def ray_project_bwd(v: torch.Tensor, kappa: torch.Tensor, active: torch.Tensor,
                    grad_y: torch.Tensor, pack_id: int, old_head: bool = False) -> torch.Tensor:
    return backward_raw(v, kappa, active, grad_y, _pack(pack_id), old_head)


def _bwd_or_detour(v, kappa, active, grad_y, pack_id, old_head):
    """The HIP backward; where the kernels decline the shape (``RAYEN_E_UNSUPPORTED``), autograd through the packed
    torch evaluator on the same device -- loudly, once per pack.  Called from the autograd formulas (NOT from inside the
    custom op: below the dispatcher's autograd key nothing would be recorded)."""
    try:
        return torch.ops.rayen_amd.ray_project_bwd(v, kappa, active, grad_y, pack_id, old_head)
    except _lib.RayenError as err:
        if err.code != _lib.E_UNSUPPORTED or os.environ.get("RAYEN_STRICT_HIP", "0") == "1":
            raise
        pack = _pack(pack_id)
        if not pack.__dict__.get("_warned_bwd"):
            warnings.warn(f"rayen_amd: no HIP backward kernel serves this constraint set ({err}); gradients come from "
                          "autograd through the packed torch evaluator (rayen_amd/eager.py) on " + str(v.device),
                          RuntimeWarning, stacklevel=2)
            pack.__dict__["_warned_bwd"] = True
        return _eager_backward(pack, v, grad_y, old_head)


def _eager_backward(pack, v, grad_y, old_head):
    from . import eager
    key = ("_eager", v.dtype)
    ev = pack.__dict__.get(key)
    if ev is None:
        ev = pack.__dict__[key] = eager.PackedEvaluator(pack.consts, v.dtype, v.device)
    with torch.enable_grad():
        leaf = v.detach().clone().requires_grad_(True)
        y, _ = ev.project(leaf, old_head=old_head)
        (grad_v,) = torch.autograd.grad(y, leaf, grad_y.to(y.dtype))
    return grad_v


@ray_project_bwd.register_fake
def _(v, kappa, active, grad_y, pack_id, old_head=False):
    return torch.empty_like(v)


def _setup_context(ctx, inputs, output):
    v, pack_id, need_active, old_head = inputs
    ctx.old_head = old_head
    if not need_active:
        raise RuntimeError("rayen_amd::ray_project was called with need_active=False on an input that "
                           "requires grad")
    _, kappa, active = output
    ctx.pack_id = pack_id
    ctx.save_for_backward(v, kappa, active)


def _backward(ctx, grad_y, grad_kappa, grad_active):
    v, kappa, active = ctx.saved_tensors
    if grad_y is None:
        return torch.zeros_like(v), None, None, None
    return (_bwd_or_detour(v, kappa, active, grad_y, ctx.pack_id, ctx.old_head), None, None, None)


ray_project.register_autograd(_backward, setup_context=_setup_context)


# ------------------------------------------------------------------------------------------------
# mapper + projection in one launch (rayen/constraint_module.py:525 followed by :468-474)
# ------------------------------------------------------------------------------------------------

def mapper_fusable(x, weight, bias, pack):
    """Can ``ray_project_mapped`` serve this call?  (fp32, contiguous rows, and a fused form for this pack and input
    width: weights read in place by the exact-fp32 family, or through their split-operand image by the default one.)"""
    if not (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and weight.dim() == 2 and weight.shape[0] == pack.consts.n and x.shape[1] == weight.shape[1]
            and weight.stride(1) == 1
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous()))):
        return False
    mode = pack.mapper_mode(weight.shape[1])
    if mode == 1:
        return weight.is_contiguous() and weight.data_ptr() % 16 == 0
    return mode == 2


@torch.library.custom_op("rayen_amd::ray_project_mapped", mutates_args=())
def ray_project_mapped(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], pack_id: int,
                       need_grad: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """``(y, kappa, active, v)`` with ``v = x weight' + bias`` evaluated inside the projection kernel.

    ``need_grad``: also write ``v`` and the active-constraint record (the backward's inputs); without it
    ``v`` never reaches memory and both come back empty."""
    pack = _pack(pack_id)
    if not mapper_fusable(x, weight, bias, pack):
        raise RuntimeError("rayen_amd::ray_project_mapped: unsupported mapper (check ops.mapper_fusable first)")
    if x.device.index != pack.device_index:
        raise RuntimeError("rayen_amd: input and constant pack live on different devices")
    if x.stride(1) != 1:
        x = x.contiguous()
    B, in_dim = x.shape
    k, n = pack.consts.k, pack.consts.n
    y = torch.empty((B, k), dtype=x.dtype, device=x.device)
    kappa = torch.empty((B,), dtype=x.dtype, device=x.device)
    active = torch.empty((B if need_grad else 0, 2), dtype=torch.int32, device=x.device)
    v = torch.empty((B if need_grad else 0, n), dtype=x.dtype, device=x.device)
    with _on_device(x.device):
        stream = _stream(x.device.index)
        if pack.mapper_mode(in_dim) == 2:
            image = pack.mapper_image(weight, bias, stream)
            code = _lib.load().rayen_ray_project_mapped_image_f32(
                pack.handle, _ptr(x), B, x.stride(0) if B else in_dim, in_dim, _ptr(image),
                _ptr(v) if need_grad else None, n, _ptr(y), k, _ptr(kappa),
                _ptr(active) if need_grad else None, _ptr(pack.nan_flag), stream)
        else:
            code = _lib.load().rayen_ray_project_mapped_f32(
                pack.handle, _ptr(x), B, x.stride(0) if B else in_dim, in_dim, _ptr(weight), weight.stride(0),
                _ptr(bias), _ptr(v) if need_grad else None, n, _ptr(y), k, _ptr(kappa),
                _ptr(active) if need_grad else None, _ptr(pack.nan_flag), stream)
    _lib.check(code, "rayen_ray_project_mapped")
    return y, kappa, active, v


@ray_project_mapped.register_fake
def _(x, weight, bias, pack_id, need_grad):
    pack = _pack(pack_id)
    B = x.shape[0]
    rows = B if need_grad else 0
    return (x.new_empty((B, pack.consts.k)), x.new_empty((B,)),
            x.new_empty((rows, 2), dtype=torch.int32), x.new_empty((rows, pack.consts.n)))


def _mapped_setup_context(ctx, inputs, output):
    x, weight, bias, pack_id, need_grad = inputs
    if not need_grad:
        raise RuntimeError("rayen_amd::ray_project_mapped was called with need_grad=False on inputs that "
                           "require grad")
    _, kappa, active, v = output
    ctx.pack_id = pack_id
    ctx.has_bias = bias is not None
    ctx.save_for_backward(x, weight, v, kappa, active)


def _mapped_backward(ctx, grad_y, grad_kappa, grad_active, grad_v_out):
    x, weight, v, kappa, active = ctx.saved_tensors
    if grad_y is None:
        return None, None, None, None, None
    # d y / d v on the HIP backward kernel; the three mapper products are plain GEMMs (rocBLAS via torch)
    grad_v = _bwd_or_detour(v, kappa, active, grad_y, ctx.pack_id, False)
    grad_x = grad_v @ weight if ctx.needs_input_grad[0] else None
    grad_w = grad_v.t() @ x if ctx.needs_input_grad[1] else None
    grad_b = grad_v.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return grad_x, grad_w, grad_b, None, None


ray_project_mapped.register_autograd(_mapped_backward, setup_context=_mapped_setup_context)
