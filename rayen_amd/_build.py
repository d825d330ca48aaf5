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
SOURCES = ["rayen_abi.hip", "rayen_generic.hip", "rayen_mfma.hip", "rayen_mfma_split.hip", "rayen_mfma_pair.hip", "rayen_mfma_pair_io.hip", "rayen_mfma_pair_wl.hip", "rayen_mfma_pair_ws8.hip", "rayen_wide.hip", "rayen_mfma_mapped.hip", "rayen_mfma_bwd.hip", "rayen_mfma_bwdg.hip", "rayen_mfma_bwdp.hip", "rayen_mfma_bwdd.hip", "rayen_lmi_wave32.hip", "rayen_lmi_wave64.hip", "rayen_lmi_block.hip", "rayen_mfma_bwdg64.hip", "rayen_mfma_bwd64.hip", "rayen_mfma_f64.hip",
           "rayen_lmi_quad32.hip", "rayen_lmi_quad64.hip"]
# On gfx950 a packed-fp32 instruction whose low result reads the HIGH half of its second source (op_sel:[0,1,..]: how hipcc's
# SLP vectoriser broadcasts the second element of a register pair) reads that operand as 0 in lanes 48-63 now and then while
# an MFMA is executing on the SIMD -- the fault behind the flat-row kernel's intermittent y == y0 of round 4 (DESIGN.md
# section 3; scripts/ubench/pkfma_hazard.hip reproduces it in 60 lines).  scripts/check_packed_opsel.py
# (tests/test_kernel_isa_hazards.py, no GPU needed) audits the ISA of EVERY kernel of the library for that form with the
# flags below; the translation units in which the vectoriser produced it are built without the vectoriser (packed arithmetic
# the kernels ask for themselves -- ext_vector_type(2) operands in their natural halves -- is unaffected), the others keep it
# (without it the fused-mapper instances of rayen_mfma_pair.hip spill 2.5 x as many registers: 0.101 against 0.086 ms).
COMMON_FLAGS = []     # extra compiler flags of every translation unit
EXTRA_FLAGS = {"rayen_generic.hip": ["-fno-slp-vectorize"],          # per-source compiler flags
               "rayen_mfma_pair_io.hip": ["-fno-slp-vectorize"],
               # (packed fp32 instructions do not issue while another wave's MFMAs execute: scripts/ubench/mfma_coissue.hip)
               "rayen_mfma_pair_wl.hip": ["-fno-slp-vectorize"],
               "rayen_mfma_bwdd.hip": ["-fno-slp-vectorize"]}
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
    """Compile every HIP source for gfx950 into ``csrc/librayen_hip.so``; returns its path.

    Translation units are compiled side by side (objects under ``csrc/_obj``, git-ignored), then linked."""
    if not force and not is_stale():
        return LIBRARY
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(CSRC, "_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-I", CSRC, *COMMON_FLAGS]

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc_path(), *flags, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + proc.stdout + proc.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        objects = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objects, "-o", LIBRARY + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + proc.stdout + proc.stderr)
    try:
        audit(LIBRARY + ".tmp")
    except Exception:
        os.remove(LIBRARY + ".tmp")
        raise
    os.replace(LIBRARY + ".tmp", LIBRARY)
    return LIBRARY


LAST_AUDIT = None      # (packed fp32 instructions, symbols) of the library audit() last passed


def audit(library=None):
    """Disassemble every gfx950 code object IN the built library and refuse one that holds a packed-fp32 instruction of the
    faulty operand form (rayen_amd/_isa_audit.py; the comment at EXTRA_FLAGS).  ``build()`` runs this on what it has just
    linked, before installing it: a compiler that vectorises differently cannot slip the form into the product.
    ``RAYEN_ALLOW_PACKED_OPSEL=1`` (experiments only) turns the refusal into a warning on stderr."""
    global LAST_AUDIT
    import sys
    from . import _isa_audit
    packed, found, symbols = _isa_audit.audit_library(library or LIBRARY)
    if found:
        kernels = sorted({k for k, _ in found})
        msg = (f"{len(found)} packed-fp32 instructions with op_sel:[0,1,..] (gfx950: src1 reads 0 in lanes 48-63 next to an MFMA) "
               f"in {len(kernels)} kernels of {library or LIBRARY}, e.g. {kernels[0]}: {found[0][1]} -- build that translation "
               "unit with -fno-slp-vectorize (rayen_amd/_build.py::EXTRA_FLAGS)")
        if os.environ.get("RAYEN_ALLOW_PACKED_OPSEL") != "1":
            raise RuntimeError("ISA audit failed: " + msg)
        print("[rayen_amd._build] WARNING: " + msg, file=sys.stderr)
    LAST_AUDIT = (packed, symbols)
    return packed, len(found), symbols


if __name__ == "__main__":
    print(build(force=True, verbose=True))
