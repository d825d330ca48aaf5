"""``ConstraintModule``: the RAYEN layer, drop-in for ``rayen.constraint_module.ConstraintModule``.

Same constructor, same ``forward`` contract (any ``[B, ...]`` tensor in, ``[B, k, 1]`` out,
rayen/constraint_module.py:520-533), same buffer names (so ``state_dict``s and
pickles written with the reference load here) and the same helper methods
(``getDimAfterMap, gety0, getyFromz, getzFromy, computeKappa``; :351, :506-518).

What differs is how ``forwardForRAYEN`` (:468-474) is executed: instead of ~165
PyTorch ops and 10 host syncs per call, one hand-written gfx950 kernel computes
``kappa`` for every constraint family and writes ``y = y0 + NA_E v / max(1, kappa(v))``
(the homogeneity identity of SURVEY.md §0, equal to :468-474 for every ``v``).
Tensors on a HIP device go to those kernels (a missing ``librayen_hip.so`` is a loud error, never a detour);
tensors on the host -- the reference's own smoke test runs there -- and constraint sets no kernel serves
(with a one-time ``RuntimeWarning``; ``RAYEN_STRICT_HIP=1`` makes it an error) are evaluated from the same packed
constants with plain torch ops on the caller's device (``rayen_amd/eager.py``; this is not the test oracle).

``method='RAYEN'`` (the default) is the hot path this package accelerates; its older step rule
``'RAYEN_old'`` (:460-466) runs on the same kernels; ``'UU'`` is the identity and kept because it
is free; the paper baselines (``UP, PP, DC3, Bar``) are out of scope and raise
``NotImplementedError``.

Documented deviation: for an SOC whose ray never meets the cone (negative
discriminant with ``c' < 0``) the reference's assert at :342 fires (or NaN under
``python -O``); here that constraint contributes ``kappa_j = 0``, its mathematical value.
"""
from __future__ import annotations

import copy
import os
import warnings

import numpy as np
import torch
import torch.nn as nn

from . import _lib, eager, ops, pack as _pack, utils


class ConstraintModule(torch.nn.Module):
    _fast = {}                    # (replaced per instance; a pickle written before round 4 has no such attribute)

    # Which (device index, dtype, old_head) combinations the HIP kernels refused (RAYEN_E_UNSUPPORTED, announced with a
    # warning): those calls -- and only those -- run rayen_amd/eager.py.  Forgotten whenever the packs are rebuilt
    # (.to(), load_state_dict) and never pickled.  ``_hip_unsupported`` is the any-of view the tests read.
    @property
    def _hip_unsupported(self):
        return bool(self.__dict__.get("_unsupported"))

    @_hip_unsupported.setter
    def _hip_unsupported(self, value):
        if value:
            self.__dict__.setdefault("_unsupported", set()).add(None)     # (None: every combination)
        else:
            self.__dict__["_unsupported"] = set()

    def _refused(self, v, old_head=False):
        refused = self.__dict__.get("_unsupported")
        return bool(refused) and (None in refused or (v.device.index, v.dtype, bool(old_head)) in refused)

    def __setattr__(self, name, value):
        super().__setattr__(name, value)
        if name == "mapper":
            self.__dict__["_fast"] = {}       # (the fast path caches "the mapper is the identity" per device and dtype)

    def __init__(self, cs, input_dim=None, method='RAYEN', create_map=True, args_DC3=None):
        super().__init__()

        self.method = method
        if method not in ('RAYEN', 'RAYEN_old', 'UU'):
            if method in ('UP', 'PP', 'DC3', 'Bar'):
                raise NotImplementedError(
                    f"method '{method}' is one of the reference's comparison baselines; rayen_amd "
                    "implements the RAYEN projection only")
            raise NotImplementedError
        self.args_DC3 = args_DC3

        self.cs = cs
        self.k = cs.k  # dimension of the ambient space
        self.n = cs.n  # dimension of the embedded space

        # every row of A_p divided by its slack at z0 (rayen/constraint_module.py:38)
        D = cs.A_p / ((cs.b_p - cs.A_p @ cs.z0) @ np.ones((1, cs.n)))

        all_P, all_q, all_r = utils.getAllPqrFromQcs(cs.qcs)
        all_M, all_s, all_c, all_d = utils.getAllMscdFromSocs(cs.socs)

        if cs.has_lmi_constraints:
            # H = F_k + sum_i y0_i F_i,  H^-1 = L L'  (:43-52); like the reference, the
            # last slot of the all_F buffer ends up holding H
            all_F = copy.deepcopy(cs.lmic.all_F)
            H = np.array(all_F[-1], dtype=np.float64)
            for i in range(cs.lmic.dim()):
                H = H + cs.y0[i, 0] * cs.lmic.all_F[i]
            all_F[-1] = H
            Hinv = np.linalg.inv(H)
            self.register_buffer("mHinv", torch.Tensor(-Hinv))
            self.register_buffer("L", torch.Tensor(np.linalg.cholesky(Hinv)))
        else:
            all_F = []

        # buffers follow .to(device) and appear in state_dict under the reference's names (:59-74)
        self.register_buffer("D", torch.Tensor(D))
        self.register_buffer("all_P", torch.Tensor(np.array(all_P)))
        self.register_buffer("all_q", torch.Tensor(np.array(all_q)))
        self.register_buffer("all_r", torch.Tensor(np.array(all_r)))
        self.register_buffer("all_M", torch.Tensor(np.array(all_M)))
        self.register_buffer("all_s", torch.Tensor(np.array(all_s)))
        self.register_buffer("all_c", torch.Tensor(np.array(all_c)))
        self.register_buffer("all_d", torch.Tensor(np.array(all_d)))
        self.register_buffer("all_F", torch.Tensor(np.array(all_F)))
        self.register_buffer("A_p", torch.Tensor(cs.A_p))
        self.register_buffer("b_p", torch.Tensor(cs.b_p))
        self.register_buffer("yp", torch.Tensor(cs.yp))
        self.register_buffer("NA_E", torch.Tensor(cs.NA_E))
        self.register_buffer("z0", torch.Tensor(cs.z0))
        self.register_buffer("y0", torch.Tensor(cs.y0))

        if cs.has_quadratic_constraints:
            # sigma, phi, delta per quadratic, evaluated at the default dtype like the reference (:99-122)
            all_delta, all_phi = [], []
            y0 = self.y0
            for i in range(self.all_P.shape[0]):
                P, q, r = self.all_P[i, :, :], self.all_q[i, :, :], self.all_r[i, :, :]
                g_y0 = 0.5 * y0.T @ P @ y0 + q.T @ y0 + r
                sigma = 2 * g_y0
                grad_row = y0.T @ P + q.T
                all_phi.append(-grad_row / sigma)
                all_delta.append((grad_row.T @ grad_row - 4 * g_y0 * 0.5 * P) / torch.square(sigma))
            self.register_buffer("all_delta", torch.stack(all_delta))
            self.register_buffer("all_phi", torch.stack(all_phi))

        if self.method == 'RAYEN':
            self.forwardForMethod = self.forwardForRAYEN
            self.dim_after_map = self.n
        elif self.method == 'RAYEN_old':
            self.forwardForMethod = self.forwardForRAYENOld
            self.dim_after_map = self.n + 1
        else:  # 'UU'
            self.forwardForMethod = self.forwardForUU
            self.dim_after_map = self.k

        if create_map:
            utils.verify(input_dim is not None, "input_dim needs to be provided")
            self.mapper = nn.Linear(input_dim, self.dim_after_map)
        else:
            self.mapper = nn.Sequential()  # mapper does nothing

        # fused NaN check: the kernel raises a device flag instead of re-reading y (:531)
        self.check_nan = True
        # evaluate the mapper inside the projection kernel when the shapes allow it (one launch, v
        # never written to memory in inference); False = always run nn.Linear as its own GEMM
        self.fuse_mapper = True
        # True: clipped samples stop 2^-20 of their step short of the boundary in fp32 instead of ON it, where half of the fp32
        # roundings fall outside (the fp32 kernels evaluate (1 + 2^-20) kappa, RAYEN_PREPARE_INWARD_BIAS in include/rayen_hip.h;
        # fp64 and interior samples untouched).  Measured on 65 536 rows of config 3 (profiles/bench/r06_inward_bias.txt): rows
        # with a positive fp64 residual 31 594 -> 1, outputs moved by at most 1.3e-6 of a row.  OFF by default: the shift is
        # 2^-20 of the STEP y - y0, and on a row whose y is small against its step (the reference's example sets: a point near the
        # origin reached from y0 = (0.5, 0.5)) that is more than the 1e-5 of the row the parity bar allows -- the golden vectors
        # fail with it on (round 6, gpurun_out/r06d).  Read when a device's constants are packed (first forward on that device).
        self.inward_bias = False
        self._device_packs = {}
        self._consts = None
        self._fast = {}

    # ------------------------------------------------------------------ constant packs
    def _invalidate_packs(self):
        self._device_packs = {}
        self._consts = None
        self._fast = {}
        self.__dict__["_unsupported"] = set()
        self.__dict__.pop("_eager", None)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate_packs()  # buffers moved or changed dtype
        return out

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._invalidate_packs()
        return out

    def packed_constants(self):
        """Host-side (fp64) row matrix + segment table derived from the buffers."""
        if self._consts is None:
            names = ("D", "NA_E", "z0", "yp", "y0", "all_phi", "all_delta", "all_M", "all_s",
                     "all_c", "all_d", "all_F", "L", "all_P")
            exact = ([(qc.P, qc.q, qc.r) for qc in self.cs.qcs], self.cs.y0) if self.cs.qcs else None
            self._consts = _pack.pack_constants({n: getattr(self, n) for n in names if hasattr(self, n)},
                                                exact_quadratics=exact)
        return self._consts

    def device_pack(self, device):
        """(pack, pack_id) with the constants resident on ``device`` (built on first use)."""
        index = device.index if device.index is not None else torch.cuda.current_device()
        entry = self._device_packs.get(index)
        if entry is None:
            # (inward_bias: RAYEN_PREPARE_INWARD_BIAS, include/rayen_hip.h -- set the attribute before the first forward)
            dp = _pack.DevicePack(self.packed_constants(), index,
                                  prepare=_lib.PREPARE_INWARD_BIAS if getattr(self, "inward_bias", False) else 0)
            entry = (dp, ops.register_pack(dp))
            self._device_packs[index] = entry
        return entry

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_device_packs"] = {}   # device handles are not picklable; rebuilt lazily
        state["_consts"] = None
        state["_fast"] = {}
        state["_unsupported"] = set()
        state.pop("_eager", None)
        state.pop("forwardForMethod", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._fast = {}
        self.forwardForMethod = {'RAYEN': self.forwardForRAYEN, 'RAYEN_old': self.forwardForRAYENOld,
                                 'UU': self.forwardForUU}[self.method]

    # ------------------------------------------------------------------ the projection
    def _project(self, q, old_head=False):
        """``q [B, >=n, 1]`` (or ``[B, >=n]``) -> ``(y [B,k], kappa [B])`` through the fused HIP op."""
        v = torch.flatten(q, 1)
        if not v.is_cuda:
            # tensors on the host (the reference's own smoke runs there, examples/test_layer.py:70-117): the packed form
            # in plain torch ops on the caller's device, differentiable through autograd -- rayen_amd/eager.py
            return eager.project(self, v, old_head=old_head)
        if v.dtype not in (torch.float32, torch.float64):
            y, kappa = self._project(v.float(), old_head=old_head)       # 16-bit activations: computed in fp32
            return y.to(v.dtype), (None if kappa is None else kappa.to(v.dtype))
        if self._refused(v, old_head):
            return eager.project(self, v, old_head=old_head)
        try:
            dp, pack_id = self.device_pack(v.device)
            need_active = torch.is_grad_enabled() and v.requires_grad
            if not need_active and type(v) is torch.Tensor and not torch.compiler.is_compiling():
                # plain inference call: straight to the C ABI (the same code the registered op runs; the dispatcher
                # layers around a custom op cost ~10 us per call, as much as the kernel at small batches)
                y, _, _ = ops.project_raw(v, dp, want_active=False, old_head=old_head, want_kappa=False)
                return y, None
            y, kappa, _ = torch.ops.rayen_amd.ray_project(v, pack_id, need_active, old_head)
            return y, kappa
        except _lib.RayenError as err:
            if err.code != _lib.E_UNSUPPORTED or os.environ.get("RAYEN_STRICT_HIP", "0") == "1":
                raise
            # a shape no HIP kernel serves: say so ONCE, then evaluate the packed form with torch ops on
            # the same device.  Never silent, never for a missing library (that raises in _lib.load), and
            # RAYEN_STRICT_HIP=1 turns it back into the error.
            warnings.warn(f"rayen_amd: no HIP kernel serves this constraint set ({err}); this module now runs the "
                          "packed torch evaluator (rayen_amd/eager.py) on " + str(v.device), RuntimeWarning, stacklevel=3)
            self.__dict__.setdefault("_unsupported", set()).add((v.device.index, v.dtype, bool(old_head)))
            return eager.project(self, v, old_head=old_head)

    def computeKappa(self, v_bar):
        """``kappa [B,1,1]`` of directions ``v_bar [B,n,1]`` (rayen/constraint_module.py:351-458)."""
        v = torch.flatten(v_bar, 1)
        if not v.is_cuda or self._refused(v) or v.dtype not in (torch.float32, torch.float64):
            return eager.evaluator_for(self, v).kappa(v[:, :self.n]).reshape(-1, 1, 1)
        dp, _ = self.device_pack(v.device)
        try:
            _, kappa, _ = ops.project_raw(v, dp, want_y=False, want_active=False)
        except _lib.RayenError as err:
            if err.code != _lib.E_UNSUPPORTED:
                raise
            # the kernels that always write y (an LMI on the workgroup-per-sample kernels, alone, next to quadratics / cones
            # or behind the products GEMM; rayen_abi.hip::mixed_forward) decline y == NULL: same kernels, a scratch y
            try:
                _, kappa, _ = ops.project_raw(v, dp, want_y=True, want_active=False)
            except _lib.RayenError as err2:
                if err2.code != _lib.E_UNSUPPORTED or os.environ.get("RAYEN_STRICT_HIP", "0") == "1":
                    raise
                return eager.evaluator_for(self, v).kappa(v[:, :self.n]).reshape(-1, 1, 1)
        return kappa.reshape(-1, 1, 1)

    def forwardForRAYEN(self, q):
        y, _ = self._project(q)
        return y.unsqueeze(2)

    def forwardForRAYENOld(self, q):
        # step 1/(exp(beta) + kappa) with beta = q[:, n] (rayen/constraint_module.py:460-466)
        y, _ = self._project(q, old_head=True)
        return y.unsqueeze(2)

    def forwardForUU(self, q):
        return q

    # ------------------------------------------------------------------ reference helper surface
    def getDimAfterMap(self):
        return self.dim_after_map

    def gety0(self):
        return self.getyFromz(self.z0)

    def getyFromz(self, z):
        return self.NA_E @ z + self.yp

    def getzFromy(self, y):
        return self.NA_E.T @ (y - self.yp)

    def _forward_fused_mapper(self, x2):
        """mapper + projection as ONE kernel launch (``rayen_amd::ray_project_mapped``) or ``None`` when
        the fused kernel does not serve this layer/input (then the two-op path below runs)."""
        if (self.method != 'RAYEN' or not getattr(self, "fuse_mapper", True)
                or not isinstance(self.mapper, nn.Linear) or not x2.is_cuda or x2.dtype != torch.float32
                or self._refused(x2)):
            return None
        try:
            dp, pack_id = self.device_pack(x2.device)
        except _lib.RayenError:
            return None           # (the two-op path below reports it)
        weight, bias = self.mapper.weight, self.mapper.bias
        if not ops.mapper_fusable(x2, weight, bias, dp):
            return None
        need_grad = torch.is_grad_enabled() and (x2.requires_grad or weight.requires_grad
                                                 or (bias is not None and bias.requires_grad))
        y, _, _, _ = torch.ops.rayen_amd.ray_project_mapped(x2, weight, bias, pack_id, need_grad)
        return y.unsqueeze(2)

    # ------------------------------------------------------------------ small-batch inference fast path
    def _fast_entry(self, x):
        """Per-(device, dtype) prebuilt call of the C ABI for plain inference with the identity mapper: pack handle,
        ctypes entry point, NaN-flag address and sizes looked up ONCE (round 4: the generic route -- flatten, the empty
        nn.Sequential, unsqueeze, _project, ops.project_raw and their re-validation -- cost ~13 us of Python per call in
        front of a 4.5 us kernel at config 1).  ``None`` when the layer has a mapper, another method, or no HIP kernel."""
        key = (x.device.index, x.dtype)
        entry = self._fast.get(key, False)
        if entry is False:
            entry = None
            if (self.method == 'RAYEN' and isinstance(self.mapper, nn.Sequential) and len(self.mapper) == 0
                    and not self._refused(x) and x.dtype in (torch.float32, torch.float64)
                    and self.n < ops._WIDE_MIN_N[x.dtype]):      # (wide sets: GEMM + products epilogue, ops.project_raw)
                try:
                    dp, _ = self.device_pack(x.device)
                    fn = ops._entry(ops._FWD[(x.dtype, False)])
                    entry = (fn, dp.handle, dp.nan_flag.data_ptr(), self.k, self.n, x.device.index, dp)
                except _lib.RayenError:
                    entry = None           # (the generic route reports it)
            self._fast[key] = entry
        return entry

    def forward(self, x):
        # x: [nsib, numel_input_mapper, 1]; after the mapper q is [nsib, numel_output_mapper, 1]
        if (type(x) is torch.Tensor and x.is_cuda and x.dim() >= 2 and x.is_contiguous()
                and not (x.requires_grad and torch.is_grad_enabled()) and not torch.compiler.is_compiling()):
            entry = self._fast_entry(x)
            if entry is not None:
                fn, handle, nan_ptr, k, n, index, dp = entry
                B = x.shape[0]
                width = x.numel() // B if B else n
                if width >= n and dp.handle is not None:
                    y = torch.empty((B, k, 1), dtype=x.dtype, device=x.device)
                    if torch.cuda.current_device() == index:
                        code = fn(handle, x.data_ptr(), B, width, y.data_ptr(), k, None, None, nan_ptr,
                                  ops._stream(index))
                    else:
                        with torch.cuda.device(index):
                            code = fn(handle, x.data_ptr(), B, width, y.data_ptr(), k, None, None, nan_ptr,
                                      ops._stream(index))
                    if code == 0:
                        if __debug__ and self.check_nan and not torch.cuda.is_current_stream_capturing():
                            if int(dp.nan_flag.item()) != 0:      # the flag read is a host sync (CM:531 is one too)
                                dp.nan_flag.zero_()
                                raise AssertionError("the projection produced NaN (NaN in the input?)")
                        return y
                    if code != _lib.E_UNSUPPORTED:
                        _lib.check(code, "rayen_ray_project")
                    self._fast[(x.device.index, x.dtype)] = None     # no kernel for this set: the route below says so
        x2 = torch.flatten(x, 1)  # == x.view(B, -1), and defined for B = 0
        y = self._forward_fused_mapper(x2)
        if y is None:
            q = torch.unsqueeze(self.mapper(x2), dim=2)
            y = self.forwardForMethod(q)

        if __debug__ and self.check_nan and self.method in ('RAYEN', 'RAYEN_old'):
            if not y.is_cuda or self._refused(y, self.method == 'RAYEN_old'):
                assert not torch.isnan(y).any(), "the projection produced NaN (NaN in the input?)"     # CM:531
            elif not torch.cuda.is_current_stream_capturing():  # the flag read is a host sync
                dp, _ = self.device_pack(y.device)
                if int(dp.nan_flag.item()) != 0:
                    dp.nan_flag.zero_()
                    raise AssertionError("the projection produced NaN (NaN in the input?)")
        return y
