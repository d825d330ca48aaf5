"""Alias of :mod:`rayen_amd.constraint_module` (see ``rayen/__init__.py``)."""
from rayen_amd.constraint_module import *  # noqa: F401,F403
