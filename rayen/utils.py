"""Alias of :mod:`rayen_amd.utils` (see ``rayen/__init__.py``)."""
from rayen_amd.utils import *  # noqa: F401,F403
