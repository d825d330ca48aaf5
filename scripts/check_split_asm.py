#!/usr/bin/env python
"""Safety check of the hand-placed loads of rayen_mfma_split.hip, rayen_mfma_pair.hip and rayen_mfma_pair_io.hip: between an asm `global_load` into a chunk of the
rolling A buffer and the `s_waitcnt` that covers it the compiler must not touch those registers (copy, spill):
it does not know the data is still in flight.  Scans the gfx950 ISA of every instance of the kernel and lists
any instruction inside the tile loops, other than the MFMAs and the loads themselves, that names a chunk register --
and any VALU write of an SGPR (a spilled SGPR restored by v_readlane_b32 ...) within the five wait states in front of
an asm vector-memory instruction that reads it (the hazard behind the device fault of the fused mapper on sets with
equality constraints, rounds 1-2: rayen_split_image.h).
    python scripts/check_split_asm.py        (exit code 1 if something is found)"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines, starts = [], []
for name in ("rayen_mfma_split", "rayen_mfma_pair", "rayen_mfma_pair_io"):
    src = os.path.join(REPO, "rayen_amd", "csrc", name + ".hip")
    asm = f"/tmp/{name}.s"
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(REPO, "include"),
                    "-I", os.path.join(REPO, "rayen_amd", "csrc"), *os.environ.get("RAYEN_CHECK_DEFS", "").split(),
                    "-S", "--cuda-device-only", src, "-o", asm],
                   check=True, stderr=subprocess.DEVNULL)
    base = len(lines)
    text = open(asm).read().split("\n")
    lines += text
    starts += [base + i for i, l in enumerate(text)
               if l.startswith(("_ZN5rayen21mfma_split_fwd_kernel", "_ZN5rayen21mfma_split_map_kernel", "_ZN5rayen20mfma_pair_fwd_kernel", "_ZN5rayen20mfma_pair_map_kernel", "_ZN5rayen19mfma_pair_io_kernel", "_ZN5rayen20mfma_pair_iof_kernel"))
               and l.split(";")[0].rstrip().endswith(":")]
    starts.append(base + len(text))          # (closes the last kernel of this file)


def regs_of(text):
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


bad_total = 0
for s, e in zip(starts[:-1], starts[1:]):
    if not lines[s].startswith("_ZN5rayen"):
        continue
    body = []
    for l in lines[s:e]:
        body.append(l)
        if "s_endpgm" in l:
            break
    found = re.search(r"ILi(\d)ELb(\d)ELb(\d)", lines[s])
    name = found.groups() if found else ("?", "?", "0")
    mapped = "split_map_kernel" in lines[s]
    nkx = re.search(r"ELb\dELb\dELi(\d)", lines[s]).group(1) if mapped else "0"
    if "pair_io_kernel" in lines[s]:                     # <NKK, TRACK>: NA_E = I only
        m2 = re.search(r"ILi(\d)ELb(\d)E", lines[s]).groups()
        name = (m2[0], m2[1], "0")
    if "pair_iof_kernel" in lines[s]:                    # <TRACK, STAGED>: n <= 32, rows stored back to back
        m2 = re.search(r"ILb(\d)ELb(\d)E", lines[s]).groups()
        name = ("1", m2[0], m2[1])
    if "pair_map_kernel" in lines[s]:                    # <NKK, TRACK, NKX, STAGED>
        m3 = re.search(r"ILi(\d)ELb(\d)ELi(\d)ELb(\d)", lines[s]).groups()
        name, nkx = (m3[0], m3[1], m3[3]), m3[2]
    # the hand-placed loads are the ones written as asm statements (the compiler's own loads of the mapper image or of
    # the rows are tracked by the compiler and need no check)
    in_asm, asm_lines = False, set()
    for i, l in enumerate(body):
        if "ASMSTART" in l:
            in_asm = True
        elif "ASMEND" in l:
            in_asm = False
        elif in_asm:
            asm_lines.add(i)
    loads = [i for i in sorted(asm_lines) if "global_load_dwordx4" in body[i] and re.search(r", s\[\d+:\d+\]", body[i])]
    # ---- hazard scan: a VALU write of an SGPR (v_readlane_b32 restoring a spilled SGPR, v_readfirstlane_b32, a VOPC /
    # carry-out into an SGPR pair) needs five wait states before a VMEM instruction reads that SGPR; hipcc pads this for
    # its own instructions only.  Every hand-placed VMEM instruction must therefore read a base that the statement
    # itself produced with an SALU copy (RAYEN_ASM_BASE_COPY); what is checked here is the emitted code: within the five
    # instructions in front of an asm VMEM instruction nothing but SALU instructions may write its scalar base.
    hazards = []
    code = [(i, body[i].split(";")[0].strip()) for i in range(len(body))]
    code = [(i, t) for i, t in code if t and not t.startswith(".") and not t.endswith(":")]
    pos = {i: n for n, (i, _) in enumerate(code)}
    for i in sorted(asm_lines):
        t = body[i].split(";")[0].strip()
        m = re.match(r"(global_load\w*|global_store\w*|buffer_\w+) .*\bs\[(\d+):(\d+)\]", t)
        if not m:
            continue
        lo, hi = int(m.group(2)), int(m.group(3))
        for back in range(1, 6):
            n = pos[i] - back
            if n < 0:
                break
            prev = code[n][1]
            w = re.match(r"(v_readlane_b32|v_readfirstlane_b32) s(\d+),", prev)
            if w and lo <= int(w.group(2)) <= hi:
                hazards.append((i, t, back, prev))
                break
            if re.match(r"s_mov_b64 s\[%d:%d\]," % (lo, hi), prev):
                break                                   # the statement's own SALU copy: what comes before it is interlocked

    chunk = set()
    for i in loads:
        chunk |= regs_of(body[i].split(",")[0])
    # in-flight windows: from the first asm load of the loop nest to the vmcnt(0) that closes it
    # the window starts at the very first hand-placed load: the initial fill of the plain instances stays in flight
    # through the first group's row loads, the fresh fetch of the mapped instances through the re-split of the
    # mapper's accumulators
    first = loads[0]
    last = max(i for i, l in enumerate(body) if "s_waitcnt vmcnt(0)" in l and i > loads[-1]) if any(
        "s_waitcnt vmcnt(0)" in l for l in body[loads[-1]:]) else len(body)
    closing = min(i for i in range(loads[-1], len(body)) if "s_waitcnt vmcnt(0)" in body[i])
    bad = []
    for i in range(first, closing):
        l = body[i].split(";")[0].strip()
        if not l or l.startswith(".") or "v_mfma" in l or i in loads:
            continue
        if regs_of(l) & chunk:
            bad.append((i, l))
    family = "pair-io flat " if "pair_iof_kernel" in lines[s] else "pair-io " if "pair_io_kernel" in lines[s] else ("pair " if "mfma_pair" in lines[s] else "")
    print(f"{family}NKK={name[0]} TRACK={name[1]} STAGED={name[2]} NKX={nkx}: {len(loads)} asm loads, chunk registers {min(chunk)}..{max(chunk)}"
          f" ({len(chunk)}), SGPR hazards in front of asm VMEM: {len(hazards)}, suspicious instructions in the loop: {len(bad)}")
    for i, t, back, prev in hazards[:6]:
        print("      hazard:", i, t[:70], "<-", back, "back:", prev)
    bad_total += len(hazards)
    for i, l in bad[:12]:
        print("     ", i, l[:110])
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
