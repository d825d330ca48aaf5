"""Build ``librayen_hip.so`` (hand-written gfx950 kernels + the C ABI) in-tree with hipcc.

The shared library is git-ignored but travels to the GPU box with the gpurun
snapshot.  ``__graft_entry__.build()`` calls :func:`build`; importing the package
never compiles anything (a missing library is a loud error in ``_lib``).
"""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
SOURCES = ["rayen_abi.hip", "rayen_generic.hip", "rayen_mfma.hip", "rayen_mfma_f64.hip"]
LIBRARY = os.environ.get("RAYEN_HIP_LIBRARY") or os.path.join(CSRC, "librayen_hip.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; cannot build librayen_hip.so")


def is_stale():
    if not os.path.exists(LIBRARY):
        return True
    built = os.path.getmtime(LIBRARY)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(INCLUDE, "rayen_hip.h"))
    return any(os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into ``csrc/librayen_hip.so``; returns its path."""
    if not force and not is_stale():
        return LIBRARY
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", INCLUDE, "-I", CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIBRARY + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    os.replace(LIBRARY + ".tmp", LIBRARY)
    return LIBRARY


if __name__ == "__main__":
    print(build(force=True, verbose=True))
