"""The split-operand kernel places its A-operand loads by hand (inline asm) because hipcc would otherwise gather
them right in front of their uses; the compiler does not see that data arrive.  This test compiles the kernel to
gfx950 ISA (no GPU needed) and checks, for every instance, that nothing but the MFMAs and the loads themselves
touches a chunk register while a load into it may be in flight, and that no hand-placed vector-memory instruction reads
a scalar base inside the wait states of a VALU write to it (scripts/check_split_asm.py)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_compiler_access_to_in_flight_chunk_registers():
    run = subprocess.run([sys.executable, os.path.join(REPO, "scripts", "check_split_asm.py")],
                         capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    lines = [l for l in run.stdout.splitlines() if l.startswith(("NKK=", "pair NKK=", "pair-io NKK=", "pair-io flat NKK="))]
    assert len(lines) == 56, run.stdout                      # bf16 triples, plain: NKK in {1, 2} x TRACK x STAGED; mapped: (NKK, NKX) in {(1,1), (2,1), (2,2)} x TRACK x STAGED; f16 pairs: the same 8 (and 8 more with 32 samples per wave) + 12; f16 pairs with LDS-trickled rows: NKK in {1, 2} x TRACK, and the flat-row form TRACK x STAGED
    assert all(l.rstrip().endswith("suspicious instructions in the loop: 0") for l in lines), run.stdout
    # ... and no asm vector-memory instruction reads a scalar base that a VALU instruction (an SGPR restored from its
    # spill lane) wrote within the five wait states in front of it
    assert all("SGPR hazards in front of asm VMEM: 0," in l for l in lines), run.stdout
