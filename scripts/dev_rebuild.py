#!/usr/bin/env python
"""Developer helper: recompile the named translation units only (flags of rayen_amd/_build.py) and relink
rayen_amd/csrc/librayen_hip.so from the objects of the last full build.  `python scripts/dev_rebuild.py rayen_mfma_pair_wl`"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayen_amd import _build  # noqa: E402

objdir = os.path.join(_build.CSRC, "_obj")
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", _build.INCLUDE, "-I", _build.CSRC, *_build.COMMON_FLAGS]
procs = []
for tu in sys.argv[1:]:
    src = tu if tu.endswith(".hip") else tu + ".hip"
    obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
    procs.append((src, subprocess.Popen([_build.hipcc_path(), *flags, *_build.EXTRA_FLAGS.get(src, []), "-c",
                                         os.path.join(_build.CSRC, src), "-o", obj], stderr=subprocess.PIPE, text=True)))
for src, p in procs:
    err = p.communicate()[1]
    if p.returncode:
        sys.exit(f"{src}:\n{err}")
objects = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in _build.SOURCES]
missing = [o for o in objects if not os.path.exists(o)]
if missing:
    sys.exit(f"objects missing (run a full build first): {missing}")
subprocess.run([_build.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objects, "-o", _build.LIBRARY], check=True)
print("relinked", _build.LIBRARY)
