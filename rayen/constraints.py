"""Alias of :mod:`rayen_amd.constraints` (see ``rayen/__init__.py``)."""
from rayen_amd.constraints import *  # noqa: F401,F403
