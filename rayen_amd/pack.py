"""Fold the layer's constant buffers into one row matrix ``W`` + a segment table.

Everything ``computeKappa`` (rayen/constraint_module.py:351-458) does before its
per-sample reductions is linear in the direction ``v``:

=========  ============================================  =================================
family     rows of ``W`` (all have ``n`` columns)          reduction (kernel epilogue)
=========  ============================================  =================================
LIN        ``D``                                          ``relu(max_i D_i v)``         CM:353
QUAD_SYM   ``phi NA_E`` ; ``G = NA_E' delta NA_E``        ``phi.v + sqrt(v'Gv)``        CM:374
QUAD_FAC   ``phi NA_E`` ; ``U`` with ``U'U = G``          ``phi.v + ||Uv||``  (low rank)
SOC        ``c'NA_E``, ``(M'beta)'NA_E`` ; ``M NA_E``     root of ``a'x^2+b'x+c'``      CM:383-399
LMI        packed lower triangle of ``-L'F_a L`` . NA_E   ``relu(lambda_max)``          CM:401-449
=========  ============================================  =================================

and the output is ``y = (NA_E z0 + yp) + NA_E v / max(1, kappa)`` (CM:468-474, 512-514).
The products with ``NA_E`` are folded in here, once, in fp64, so the kernels never
form ``rho = NA_E v``.  The constants are taken from the module's *buffers* (the
same tensors the reference reads in its forward), upcast to fp64.

Row-count reductions that keep the value of every reduction unchanged:
all-zero ``D`` rows are dropped (``relu`` makes them no-ops); an SOC block ``M NA_E``
with more rows than columns is replaced by its triangular QR factor (same
``||M NA_E v||``); a quadratic that is EXACTLY low rank (rank ``<= n/2``) is stored as
the factor ``U`` (rank rows instead of n), which also makes its radicand a sum of
squares.  "Exactly" is decided on the fp64 constraint data the module was built from
(``exact_quadratics``), at the fp64 noise floor: the eigenvalues of the fp32 *buffer*
that are dropped are then provably rounding noise of the buffer, never curvature
(a form like ``diag(1, 1e-6, 1e-6, 1e-6)`` keeps its full rank and the dense layout).
Without that data only the fp64 noise floor of the buffer itself is dropped.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np

from . import _lib


@dataclass
class Segment:
    type: int
    row0: int
    nrows: int
    aux_row: int = -1
    dim: int = 0
    f0: float = 0.0
    f1: float = 0.0


@dataclass
class PackedConstants:
    """Host-side (fp64) description handed to ``rayen_pack_create``."""
    k: int
    n: int
    W: np.ndarray                      # [n_rows, n]
    segments: list = field(default_factory=list)
    NA_E: np.ndarray = None            # [k, n]
    y0: np.ndarray = None              # [k]  (= NA_E z0 + yp)
    out_identity: bool = False
    dropped_segments: int = 0          # quadratics that vanish identically in the subspace and were left out (kappa = 0)

    @property
    def n_rows(self):
        return self.W.shape[0]


def _f64(t):
    if t is None:
        return None
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


def _buffer_eps(t):
    if hasattr(t, "dtype") and str(t.dtype).endswith("float64"):
        return np.finfo(np.float64).eps
    return np.finfo(np.float32).eps


def _true_rank(P, q, r, y0, N):
    """Numerical rank (fp64 noise floor) of ``N' delta N`` for the quadratic ``(P, q, r)`` at ``y0``,
    with ``delta`` as in rayen/constraint_module.py:99-122 evaluated in fp64."""
    y0 = y0.reshape(-1, 1)
    g = P @ y0 + q.reshape(-1, 1)
    sigma = 2.0 * float((0.5 * y0.T @ P @ y0 + q.reshape(1, -1) @ y0 + r).item())
    delta = (g @ g.T - sigma * P) / (sigma * sigma)
    G = N.T @ delta @ N
    lam = np.linalg.eigvalsh(0.5 * (G + G.T))
    lam_max = max(float(lam[-1]), 0.0)
    return int(np.count_nonzero(lam > 64.0 * N.shape[1] * np.finfo(np.float64).eps * lam_max)), lam_max


def pack_constants(buffers: dict, low_rank: bool = True, exact_quadratics=None) -> PackedConstants:
    """``buffers``: the module's ``D, NA_E, z0, yp, y0, all_phi, all_delta, all_M, all_s, all_c,
    all_d, all_F, L`` (+ ``all_P`` when ``exact_quadratics`` is given; torch tensors or arrays;
    absent/empty families may be missing).

    ``exact_quadratics``: ``(list of (P, q, r) in fp64, y0 in fp64)`` -- the constraint data the
    buffers were rounded from.  It is trusted for a quadratic only if its ``P`` rounds to exactly the
    module's ``all_P`` buffer (a ``state_dict`` loaded from elsewhere fails that and falls back)."""
    eps = _buffer_eps(buffers["D"])
    D = _f64(buffers["D"])
    N = _f64(buffers["NA_E"])
    k, n = N.shape
    y0 = _f64(buffers["y0"]).reshape(k)
    z0 = _f64(buffers["z0"]).reshape(n)
    yp = _f64(buffers["yp"]).reshape(k)

    rows, segments = [], []
    dropped = 0

    def add_rows(block):
        start = sum(r.shape[0] for r in rows)
        rows.append(np.asarray(block, dtype=np.float64).reshape(-1, n))
        return start

    # ---- linear (CM:38, CM:353)
    keep = np.any(D != 0.0, axis=1)
    if np.any(keep):
        row0 = add_rows(D[keep])
        segments.append(Segment(_lib.SEG_LIN, row0, int(np.count_nonzero(keep))))

    # ---- convex quadratic (CM:99-122, CM:374)
    phi = _f64(buffers.get("all_phi"))
    delta = _f64(buffers.get("all_delta"))
    P_buf = buffers.get("all_P")

    def _exact_rank(i):
        if exact_quadratics is None or P_buf is None:
            return None
        triples, y0_exact = exact_quadratics
        if i >= len(triples):
            return None
        P, q, r = (np.asarray(a, dtype=np.float64) for a in triples[i])
        Pb = P_buf[i].detach().cpu().numpy() if hasattr(P_buf, "detach") else np.asarray(P_buf[i])
        if Pb.shape != P.shape or not np.array_equal(P.astype(Pb.dtype), Pb):
            return None
        return _true_rank(P, q, r, np.asarray(y0_exact, dtype=np.float64), N)[0]

    def _vanishes_in_subspace(i):
        """A quadratic that does not depend on z at all: P N = 0 and (P y0 + q)' N = 0 in the EXACT (fp64) constraint
        data, relative to the data's own size -- e.g. ``||velocity control point||^2 <= v_max^2`` for a control point
        the equality constraints pin to zero (config 5: 8 of 72).  Its kappa is identically 0 (CM:374 evaluates
        rounding noise of 1e-18 there, and -- at fp32 -- NaN from a negative radicand): the segment is dropped, as
        all-zero rows of D are."""
        if exact_quadratics is None or P_buf is None:
            return False
        triples, y0_exact = exact_quadratics
        if i >= len(triples):
            return False
        P, q, r = (np.asarray(a, dtype=np.float64) for a in triples[i])
        Pb = P_buf[i].detach().cpu().numpy() if hasattr(P_buf, "detach") else np.asarray(P_buf[i])
        if Pb.shape != P.shape or not np.array_equal(P.astype(Pb.dtype), Pb):
            return False
        g = P @ np.asarray(y0_exact, dtype=np.float64).reshape(-1, 1) + q.reshape(-1, 1)
        size_p, size_g = float(np.abs(P).max()), float(np.abs(g).max())
        return (float(np.abs(P @ N).max()) <= 1e-12 * max(size_p, 1e-300)
                and float(np.abs(g.T @ N).max()) <= 1e-12 * max(size_g, size_p, 1e-300))

    if phi is not None and phi.ndim == 3:
        for i in range(phi.shape[0]):
            if _vanishes_in_subspace(i):
                dropped += 1
                continue
            aux = add_rows(phi[i].reshape(1, k) @ N)
            G = N.T @ delta[i] @ N
            G = 0.5 * (G + G.T)
            lam, vec = np.linalg.eigh(G)
            lam_max = max(float(lam[-1]), 0.0)
            rank = n
            if low_rank and lam_max > 0.0:
                rank = _exact_rank(i)
                if rank is None:   # no trusted fp64 data: drop the fp64 noise floor of the buffer only
                    rank = int(np.count_nonzero(lam > 64.0 * n * np.finfo(np.float64).eps * lam_max))
                # the eigenvalues kept must stand clear of the buffer's own rounding noise
                if rank <= n // 2 and not (rank > 0 and lam[n - rank] > 16.0 * eps * lam_max):
                    rank = n
            if rank <= n // 2:
                big = np.zeros(n, dtype=bool)
                big[n - rank:] = True
                U = (np.sqrt(lam[big])[:, None]) * vec[:, big].T
                row0 = add_rows(U)
                segments.append(Segment(_lib.SEG_QUAD_FAC, row0, rank, aux_row=aux))
            else:
                row0 = add_rows(G)
                segments.append(Segment(_lib.SEG_QUAD_SYM, row0, n, aux_row=aux))

    # ---- second-order cone (CM:383-399)
    M_all = _f64(buffers.get("all_M"))
    if M_all is not None and M_all.ndim == 3:
        s_all, c_all, d_all = _f64(buffers["all_s"]), _f64(buffers["all_c"]), _f64(buffers["all_d"])
        y0c = y0.reshape(k, 1)
        for j in range(M_all.shape[0]):
            M, s, c, d = M_all[j], s_all[j], c_all[j], d_all[j]
            beta = M @ y0c + s
            tau = float((c.T @ y0c + d).item())
            MN = M @ N
            aux = add_rows(np.concatenate((c.T @ N, beta.T @ MN), axis=0))
            if MN.shape[0] > n:
                MN = np.linalg.qr(MN, mode="r")
            row0 = add_rows(MN)
            segments.append(Segment(_lib.SEG_SOC, row0, MN.shape[0], aux_row=aux,
                                    f0=tau, f1=float((beta.T @ beta).item()) - tau * tau))

    # ---- LMI (CM:43-52, CM:401-449)
    F = _f64(buffers.get("all_F"))
    if F is not None and F.ndim == 3:
        L = _f64(buffers["L"])
        r = L.shape[0]
        # (BLAS products: as einsum contractions these two took 84 s on the host for k = 1000, r = 100 -- the LMI sweep's
        # 50 s of `setup_s` in round 3)
        G_a = -np.matmul(L.T[None, :, :], np.matmul(F[:-1], L))  # [k, r, r], symmetric
        if k == n and np.array_equal(N, np.eye(k)):
            G_b = G_a
        else:
            G_b = (N.T @ G_a.reshape(k, r * r)).reshape(n, r, r)  # fold NA_E: [n, r, r]
        G_b = 0.5 * (G_b + np.transpose(G_b, (0, 2, 1)))
        il, jl = np.tril_indices(r)                              # packed index p(p+1)/2 + q, p >= q
        row0 = add_rows(G_b[:, il, jl].T)
        segments.append(Segment(_lib.SEG_LMI, row0, len(il), dim=r))

    W = np.concatenate(rows, axis=0) if rows else np.zeros((0, n))
    identity = (k == n) and np.array_equal(N, np.eye(k))
    return PackedConstants(k=k, n=n, W=np.ascontiguousarray(W), segments=segments,
                           NA_E=np.ascontiguousarray(N), y0=N @ z0 + yp, out_identity=identity,
                           dropped_segments=dropped)


def _as_double_ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class DevicePack:
    """Owner of one ``RayenPack*`` (constants resident on one HIP device)."""

    def __init__(self, consts: PackedConstants, device_index: int, prepare: int = 0, fp32_mode: int = 0):
        """``rayen_pack_create`` builds every device image and measures the fp32 kernel families here, once:
        calls on the pack never allocate, and it can be captured into a HIP graph from its first call."""
        import torch
        self.consts = consts
        self.device_index = int(device_index)
        lib = _lib.load()
        segs = (_lib.RayenSegment * max(1, len(consts.segments)))()
        for i, s in enumerate(consts.segments):
            segs[i] = _lib.RayenSegment(s.type, s.row0, s.nrows, s.aux_row, s.dim, 0, s.f0, s.f1)
        W = np.ascontiguousarray(consts.W, dtype=np.float64)
        N = np.ascontiguousarray(consts.NA_E, dtype=np.float64)
        y0 = np.ascontiguousarray(consts.y0, dtype=np.float64)
        desc = _lib.RayenPackDesc(_lib.ABI_VERSION, consts.k, consts.n, W.shape[0],
                                  len(consts.segments), int(consts.out_identity),
                                  _as_double_ptr(W), segs, _as_double_ptr(N), _as_double_ptr(y0),
                                  int(prepare), int(fp32_mode))
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device_index):
            _lib.check(lib.rayen_pack_create(ctypes.byref(desc), ctypes.byref(handle)), "rayen_pack_create")
            self.nan_flag = torch.zeros(1, dtype=torch.int32, device=f"cuda:{self.device_index}")
        self.handle = handle

    def info(self):
        out = _lib.RayenPackInfo()
        _lib.check(_lib.load().rayen_pack_info(self.handle, ctypes.byref(out)), "rayen_pack_info")
        return out

    def products_matrix(self, dtype):
        """``W_ext' [n, rows]`` on the device at ``dtype`` (``W_ext = [W ; NA_E]``, ``NA_E`` only for sets with equality
        constraints) for the wide route -- ``T = v W_ext'`` by the vendor GEMM, then ``rayen_ray_project_from_products_*``
        (include/rayen_hip.h, ABI v7).  ``None`` when the pack has no such route (an LMI segment).  Built once per dtype."""
        import torch
        cache = self.__dict__.setdefault("_products", {})
        if dtype not in cache:
            rows = int(_lib.load().rayen_products_rows(self.handle))
            if rows <= 0 or not _lib.load().rayen_products_served(self.handle, int(dtype == torch.float64)):
                cache[dtype] = None
            else:
                c = self.consts
                W_ext = c.W if c.out_identity else np.concatenate((c.W, c.NA_E), axis=0)
                assert W_ext.shape[0] == rows, (W_ext.shape, rows)
                cache[dtype] = torch.as_tensor(np.ascontiguousarray(W_ext.T), dtype=torch.float64).to(
                    device=f"cuda:{self.device_index}", dtype=dtype).contiguous()
        return cache[dtype]

    def mapper_mode(self, in_dim):
        """``rayen_mapper_fusable``: 0 = no fused form | 1 = weights read in place (exact-fp32 family) |
        2 = through a split-operand image of the weights (``rayen_mapper_prepare_f32``)."""
        import torch
        cache = self.__dict__.setdefault("_fusable", {})   # (the answer is fixed at pack creation)
        if in_dim not in cache:
            with torch.cuda.device(self.device_index):
                cache[in_dim] = int(_lib.load().rayen_mapper_fusable(self.handle, int(in_dim)))
        return cache[in_dim]

    def mapper_fusable(self, in_dim):
        """True when one of the fused mapper + projection entry points serves an ``in_dim``-wide mapper."""
        return self.mapper_mode(in_dim) != 0

    def mapper_image(self, weight, bias, stream):
        """The split-operand image of ``(weight, bias)`` for ``mapper_mode == 2``.

        Parameters that are being trained (``requires_grad``) get a fresh image on EVERY call -- one 8-block launch on
        ``stream`` -- because an optimiser may update them through ``.data`` (``p.data.add_()``, EMA code), which the
        tensor's version counter does not see.  Frozen parameters are cached per pack: the key is the storage address,
        the version counter, the shape and the stream the image was built on (a call from another stream rebuilds it
        rather than reading an image whose conversion may still be in flight there).  Weights of a frozen module that
        are nevertheless rewritten through ``.data`` need :meth:`invalidate_mapper_image`."""
        import torch
        trained = weight.requires_grad or (bias is not None and bias.requires_grad)
        key = (weight.data_ptr(), weight._version, tuple(weight.shape), weight.stride(0),
               None if bias is None else (bias.data_ptr(), bias._version), getattr(stream, "value", stream))
        cached = self.__dict__.get("_mapper_image")
        # (while a stream is being captured the conversion launch must be part of the graph: replays follow the weights)
        if (cached is not None and cached[0] == key and not trained
                and not torch.cuda.is_current_stream_capturing()):
            return cached[1]
        lib = _lib.load()
        in_dim = weight.shape[1]
        nbytes = int(lib.rayen_mapper_image_bytes(self.handle, int(in_dim)))
        if nbytes <= 0:
            raise RuntimeError("rayen_amd: this pack has no image-based fused mapper for this input width")
        image = cached[1] if (cached is not None and cached[1].numel() == nbytes) else \
            torch.empty(nbytes, dtype=torch.uint8, device=f"cuda:{self.device_index}")
        with torch.cuda.device(self.device_index):
            _lib.check(lib.rayen_mapper_prepare_f32(self.handle, ctypes.c_void_p(weight.data_ptr()), weight.stride(0),
                                                    int(in_dim), None if bias is None else ctypes.c_void_p(bias.data_ptr()),
                                                    ctypes.c_void_p(image.data_ptr()), stream), "rayen_mapper_prepare_f32")
        self.__dict__["_mapper_image"] = (key, image)
        return image

    def invalidate_mapper_image(self):
        """Forget the cached mapper image (frozen weights rewritten behind the version counter, e.g. ``p.data.copy_``)."""
        cached = self.__dict__.get("_mapper_image")
        if cached is not None:
            self.__dict__["_mapper_image"] = (None, cached[1])      # keep the buffer, drop the key

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().rayen_pack_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass
