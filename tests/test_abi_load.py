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
    for code in range(-8, 0):
        assert "unknown" not in _lib.strerror(code)
    assert "unknown" in _lib.strerror(-99)
    # argument validation happens before any device call
    assert lib.rayen_pack_create(None, None) == -1
    info = _lib.RayenPackInfo()
    assert lib.rayen_pack_info(None, ctypes.byref(info)) == -1
    lib.rayen_pack_destroy(None)


def test_struct_layouts_match_the_header():
    assert ctypes.sizeof(_lib.RayenSegment) == 40
    assert ctypes.sizeof(_lib.RayenPackInfo) == 88    # (ABI v5: + the two backward check values)
    assert ctypes.sizeof(_lib.RayenPackDesc) == 32 + 4 * ctypes.sizeof(ctypes.c_void_p)
    bad = _lib.RayenPackDesc()
    bad.abi_version = 999
    handle = ctypes.c_void_p()
    assert _lib.load().rayen_pack_create(ctypes.byref(bad), ctypes.byref(handle)) == -2   # RAYEN_E_ABI


def test_header_is_plain_c_and_links_against_the_library(tmp_path):
    """A C translation unit (gcc, -std=c99 -pedantic) includes the header, takes the address of every
    entry point and links against librayen_hip.so: what a cgo / JNI / ctypes-free binding would do."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    lib_path = _build.build()
    src = tmp_path / "abi_probe.c"
    body = "\n".join(f"  table[n++] = (void (*)(void)){name};" for name in _declared_symbols())
    src.write_text(f"""
#include <stdio.h>
#include "rayen_hip.h"
int main(void) {{
  void (*table[64])(void);
  int n = 0;
{body}
  RayenPackDesc desc;
  RayenPack* pack = NULL;
  desc.abi_version = 999;
  if (rayen_abi_version() != RAYEN_ABI_VERSION) return 2;
  if (rayen_pack_create(&desc, &pack) != RAYEN_E_ABI) return 3;
  if (sizeof(RayenSegment) != 40 || sizeof(RayenPackInfo) != 88 || sizeof(RayenPackDesc) != 64) return 4;
  for (int i = 0; i < n; ++i) if (table[i] == NULL) return 5;
  printf("%d %s\\n", n, rayen_strerror(RAYEN_E_UNSUPPORTED));
  return 0;
}}
""")
    exe = tmp_path / "abi_probe"
    libdir = os.path.dirname(lib_path)
    cmd = [gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), str(src),
           "-L", libdir, "-l:" + os.path.basename(lib_path), "-Wl,-rpath," + libdir, "-o", str(exe)]
    built = subprocess.run(cmd, capture_output=True, text=True)
    assert built.returncode == 0, built.stderr
    ran = subprocess.run([str(exe)], capture_output=True, text=True)
    assert ran.returncode == 0, (ran.returncode, ran.stdout, ran.stderr)
    assert ran.stdout.split()[0] == str(len(_declared_symbols()))
