"""The C-ABI library builds, loads and exports every symbol include/rayen_hip.h declares (no GPU)."""
import ctypes
import os
import re

from rayen_amd import _build, _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "rayen_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rayen_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_the_header():
    path = _build.build()
    assert os.path.exists(path)
    lib = _lib.load()
    declared = _declared_symbols()
    assert set(declared) == set(_lib.EXPORTS), "binding and header disagree"
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rayen_abi_version() == _lib.ABI_VERSION


def test_error_strings_and_null_handling():
    lib = _lib.load()
    assert _lib.strerror(0) == "ok"
    for code in range(-7, 0):
        assert "unknown" not in _lib.strerror(code)
    assert "unknown" in _lib.strerror(-99)
    # argument validation happens before any device call
    assert lib.rayen_pack_create(None, None) == -1
    info = _lib.RayenPackInfo()
    assert lib.rayen_pack_info(None, ctypes.byref(info)) == -1
    lib.rayen_pack_destroy(None)


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_lib.RayenSegment) == 40
    assert ctypes.sizeof(_lib.RayenPackInfo) == 40
    assert ctypes.sizeof(_lib.RayenPackDesc) == 24 + 4 * ctypes.sizeof(ctypes.c_void_p)
    bad = _lib.RayenPackDesc()
    bad.abi_version = 999
    handle = ctypes.c_void_p()
    assert _lib.load().rayen_pack_create(ctypes.byref(bad), ctypes.byref(handle)) == -2   # RAYEN_E_ABI
