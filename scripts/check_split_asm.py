#!/usr/bin/env python
"""Safety check of the hand-placed loads of rayen_mfma_split.hip, rayen_mfma_pair.hip and rayen_mfma_pair_io.hip: between an asm `global_load` into a chunk of the
rolling A buffer and the `s_waitcnt` that covers it the compiler must not touch those registers (copy, spill):
it does not know the data is still in flight.  Scans the gfx950 ISA of every instance of the kernel and lists
any instruction inside the tile loops, other than the MFMAs and the loads themselves, that names a chunk register.
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
               if l.startswith(("_ZN5rayen21mfma_split_fwd_kernel", "_ZN5rayen21mfma_split_map_kernel", "_ZN5rayen20mfma_pair_fwd_kernel", "_ZN5rayen20mfma_pair_map_kernel", "_ZN5rayen19mfma_pair_io_kernel"))
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
    if "pair_map_kernel" in lines[s]:                    # <NKK, TRACK, NKX>: never staged
        m3 = re.search(r"ILi(\d)ELb(\d)ELi(\d)", lines[s]).groups()
        name, nkx = (m3[0], m3[1], "0"), m3[2]
    # the hand-placed loads are the ones written as asm statements (the compiler's own loads of the mapper image or of
    # the rows are tracked by the compiler and need no check)
    loads = [i for i, l in enumerate(body) if "global_load_dwordx4" in l and re.search(r", s\[\d+:\d+\]", l)
             and i > 0 and "ASMSTART" in body[i - 1]]
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
    family = "pair-io " if "pair_io_kernel" in lines[s] else ("pair " if "mfma_pair" in lines[s] else "")
    print(f"{family}NKK={name[0]} TRACK={name[1]} STAGED={name[2]} NKX={nkx}: {len(loads)} asm loads, chunk registers {min(chunk)}..{max(chunk)}"
          f" ({len(chunk)}), suspicious instructions in the loop: {len(bad)}")
    for i, l in bad[:12]:
        print("     ", i, l[:110])
    bad_total += len(bad)
sys.exit(1 if bad_total else 0)
