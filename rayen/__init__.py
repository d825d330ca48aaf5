"""Import shim: ``from rayen import constraints, constraint_module`` resolves to the MI355X build.

Code written against leggedrobotics/rayen keeps its imports unchanged when this repository (instead
of the reference) is on ``sys.path``.  Everything lives in :mod:`rayen_amd`.
"""
from rayen_amd import constraint_module, constraints, utils  # noqa: F401

__all__ = ["constraints", "constraint_module", "utils"]
