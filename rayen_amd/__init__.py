"""rayen_amd -- MI355X-native RAYEN projection layer.

Drop-in for the hot path of leggedrobotics/rayen: ``ConstraintModule.forward`` with
``method='RAYEN'`` behind the reference's own Python API
(``constraints.ConvexConstraints`` + the ``torch.nn.Module`` ``ConstraintModule``).

    from rayen_amd import constraints, constraint_module
"""
from . import constraints, utils  # noqa: F401
from . import constraint_module  # noqa: F401

__all__ = ["constraints", "constraint_module", "utils"]
__version__ = "0.1.0"
