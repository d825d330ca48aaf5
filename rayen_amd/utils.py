"""Small host-side helpers that the constraint description and the layer share.

Mirrors the helper surface of the reference's ``rayen/utils.py`` that is on the
RAYEN path and that this package itself calls (``verify`` utils.py:21-23,
``getAllPqrFromQcs`` / ``getAllMscdFromSocs`` utils.py:25-46, the symmetry / non-zero
checks utils.py:113-121, ``all_equal`` utils.py:245-251).  Everything else in the
reference's utils (timers, pycddlib ``H_to_V``, ``rref``, power-iteration
experiments, pickle helpers) serves the harness or the paper baselines and is
out of scope.
"""
from __future__ import annotations

import numpy as np

_BOLD = "\033[1m"
_RESET = "\033[0m"
_COLOURS = {"blue": "\033[34m", "green": "\033[32m"}


def _print_bold(colour: str, text: str) -> None:
    print(f"{_BOLD}{_COLOURS[colour]}{text}{_RESET}")


def printInBoldBlue(data_string):
    _print_bold("blue", data_string)


def printInBoldGreen(data_string):
    _print_bold("green", data_string)


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


def all_equal(iterator):
    items = list(iterator)
    return all(item == items[0] for item in items[1:])
