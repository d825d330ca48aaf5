"""fp64 numpy evaluation of the packed form (W + segment table) -- test infrastructure.

This is the executable specification of what the HIP kernels compute from the
constants ``rayen_amd.pack.pack_constants`` produces.  Running it on the CPU and
comparing with the oracle checks the host-side folding (NA_E products, low-rank
factors, QR of SOC blocks, packed LMI generators) without a GPU.
"""
import numpy as np

from rayen_amd import _lib


def evaluate(consts, v):
    """``v [B,n]`` -> (y [B,k], kappa [B], active [B]) in fp64."""
    v = np.asarray(v, dtype=np.float64)
    T = v @ consts.W.T                                   # [B, n_rows]
    B = v.shape[0]
    kappa = np.zeros(B)
    active = np.full(B, -1)
    for si, s in enumerate(consts.segments):
        main = T[:, s.row0:s.row0 + s.nrows]
        if s.type == _lib.SEG_LIN:
            val = np.max(main, axis=1)
        elif s.type == _lib.SEG_QUAD_SYM:
            val = T[:, s.aux_row] + np.sqrt(np.maximum(np.sum(main * v, axis=1), 0.0))
        elif s.type == _lib.SEG_QUAD_FAC:
            val = T[:, s.aux_row] + np.sqrt(np.sum(main * main, axis=1))
        elif s.type == _lib.SEG_SOC:
            cr, br = T[:, s.aux_row], T[:, s.aux_row + 1]
            cp = np.sum(main * main, axis=1) - cr * cr
            bp = 2 * br - 2 * cr * s.f0
            disc = bp * bp - 4 * s.f1 * cp
            root = np.sqrt(np.maximum(disc, 0.0))
            val = np.maximum((-bp - root) / (2 * s.f1), (-bp + root) / (2 * s.f1))
            val = np.where(disc >= 0, val, 0.0)
        elif s.type == _lib.SEG_LMI:
            r = s.dim
            il, jl = np.tril_indices(r)
            A = np.zeros((B, r, r))
            A[:, il, jl] = main
            A[:, jl, il] = main
            val = np.linalg.eigvalsh(A)[:, -1]
        else:
            raise ValueError(s.type)
        better = val > kappa
        kappa = np.where(better, val, kappa)
        active = np.where(better, si, active)
    y = consts.y0[None, :] + (v @ consts.NA_E.T) / np.maximum(1.0, kappa)[:, None]
    return y, kappa, active
