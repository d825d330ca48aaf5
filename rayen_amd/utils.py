"""Small host-side helpers that the constraint description and the layer share.

Mirrors the helper surface of the reference's ``rayen/utils.py`` that is on the
RAYEN path (``verify`` utils.py:21-23, ``getAllPqrFromQcs`` / ``getAllMscdFromSocs``
utils.py:25-46, the matrix checks utils.py:113-130, ``quadExpression``
utils.py:228-242, ``all_equal`` utils.py:245-251, ``CudaTimer`` utils.py:49-61)
so that user code written against the reference keeps working.  Everything
that only serves the paper baselines (pycddlib ``H_to_V``, ``rref``, power
iteration experiments, pickle helpers) is out of scope.
"""
from __future__ import annotations

import numpy as np
import torch

_BOLD = "\033[1m"
_RESET = "\033[0m"
_COLOURS = {"blue": "\033[34m", "red": "\033[31m", "green": "\033[32m", "white": "\033[37m"}


def _print_bold(colour: str, text: str) -> None:
    print(f"{_BOLD}{_COLOURS[colour]}{text}{_RESET}")


def printInBoldBlue(data_string):
    _print_bold("blue", data_string)


def printInBoldRed(data_string):
    _print_bold("red", data_string)


def printInBoldGreen(data_string):
    _print_bold("green", data_string)


def printInBoldWhite(data_string):
    _print_bold("white", data_string)


def verify(condition, message="Condition not satisfied"):
    """Raise ``RuntimeError(message)`` when ``condition`` is false (utils.py:21-23)."""
    if not bool(condition):
        raise RuntimeError(message)


def getAllPqrFromQcs(qcs):
    """Split a list of quadratic constraints into three parallel lists (utils.py:25-33)."""
    return [qc.P for qc in qcs], [qc.q for qc in qcs], [qc.r for qc in qcs]


def getAllMscdFromSocs(socs):
    """Split a list of SOC constraints into four parallel lists (utils.py:35-46)."""
    return ([soc.M for soc in socs], [soc.s for soc in socs],
            [soc.c for soc in socs], [soc.d for soc in socs])


def isZero(A):
    return not np.any(A)


def checkMatrixisNotZero(A):
    verify(not isZero(A))


def checkMatrixisSymmetric(A):
    verify(A.shape[0] == A.shape[1])
    verify(np.allclose(A, A.T))


def checkMatrixisPsd(A, tol=0.0):
    checkMatrixisSymmetric(A)
    eigenvalues = np.linalg.eigvals(A)
    verify(np.all(eigenvalues >= -tol), f"Matrix is not PSD, min eigenvalue is {np.amin(eigenvalues)}")


def checkMatrixisPd(A):
    checkMatrixisSymmetric(A)
    eigenvalues = np.linalg.eigvals(A)
    verify(np.all(eigenvalues > 0.0), f"Matrix is not PD, min eigenvalue is {np.amin(eigenvalues)}")


def all_equal(iterator):
    items = list(iterator)
    return all(item == items[0] for item in items[1:])


def quadExpression(y, P, q, r):
    """Batched ``0.5 y'Py + q'y + r`` for ``y [B,k,1]`` (utils.py:228-242)."""
    P = P.to(y.device)
    q = q.to(y.device)
    r = r.to(y.device)
    if q.ndim == 2:
        qT = q.T
    else:
        assert q.ndim == 3
        qT = torch.transpose(q, 1, 2)
    result = 0.5 * torch.transpose(y, 1, 2) @ P @ y + qT @ y + r
    assert result.shape == (y.shape[0], 1, 1)
    return result


class CudaTimer:
    """Pair of device events around a region, seconds out (utils.py:49-61).

    On ROCm ``torch.cuda.Event`` is a HIP event recorded on torch's current
    stream, which is also the stream the fused projection is launched on.
    """

    def start(self):
        self.start_event = torch.cuda.Event(enable_timing=True)
        self.end_event = torch.cuda.Event(enable_timing=True)
        self.start_event.record()

    def endAndGetTimeSeconds(self):
        self.end_event.record()
        torch.cuda.synchronize()
        return 1e-3 * self.start_event.elapsed_time(self.end_event)
