#!/usr/bin/env python3
"""ISA audit of rayen_mfma_pair_ws.hip (no GPU needed; run by tests/test_ws_kernel_isa.py).

The kernel keeps its MFMA accumulators in v[192:255] (and the epilogues' running values in v[184:189]) BY NAME inside
asm statements and is compiled with amdgpu_num_vgpr(184); hipcc pads no hazard of an asm statement.  This script compiles the translation unit with
-save-temps and checks, for every instance of the kernel:
  1. no instruction outside ;;#ASMSTART / ;;#ASMEND names a register of v[184:255] (the named VGPRs) or a[192:255] (the
     A operands of a wave's first two tiles, also named literally);
  2. no scratch memory, no spilled VGPRs;
  3. the B operand of an asm MFMA is not written by a VALU instruction in the two instructions in front of it
     (VALU write -> MFMA source read needs wait states hipcc does not insert for an asm statement), except where the
     statement itself opens with s_nop 1;
  4. an accumulator register is read (v_max3 / v_fma / ds_write inside asm) at least MIN_GAP MFMAs after the last MFMA
     that wrote it, or behind s_nop's of >= 12 wait states.
Exit code 0 = clean."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rayen_amd", "csrc")
MIN_GAP = 2
ACC = set(range(192, 256))          # accumulators (VGPRs) | named tiles of W (accumulator registers)
NAMED_V = set(range(184, 256))      # + the epilogues' running values


def aregs_of(text):
    out = set()
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\ba(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def regs_of(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv\[(\d+)\]", text):
        out.add(int(m.group(1)))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def compile_to_asm(workdir):
    src = os.path.join(CSRC, "rayen_mfma_pair_ws.hip")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wno-inline-asm",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", src, "-o", os.path.join(workdir, "ws.o"), "-save-temps=obj"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return os.path.join(workdir, f)
    raise RuntimeError("no device assembly produced")


def kernels(path):
    cur, name = None, None
    for line in open(path).read().split("\n"):
        m = re.match(r"^(_ZN5rayen19mfma_pair_ws_kernel\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None


def audit(name, lines):
    problems = []
    in_asm = False
    instrs = []      # (text, inside an asm statement)
    for line in lines:
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if s:
            instrs.append((s, in_asm))
    last_write = {}      # accumulator register -> count of MFMAs at its last write
    n_mfma = 0
    wait_states = 0      # s_nop wait states since the last MFMA
    for idx, (s, inside) in enumerate(instrs):
        op = s.split()[0]
        if not inside and (regs_of(s) & NAMED_V):
            problems.append(f"{name}: compiler instruction names a register of v[184:255]: {s}")
        if not inside and (aregs_of(s) & ACC):
            problems.append(f"{name}: compiler instruction names a[192:255] (the named tiles of W): {s}")
        if op.startswith("scratch_"):
            problems.append(f"{name}: scratch access: {s}")
        if op == "s_nop":
            wait_states += int(s.split()[1]) + 1
        if op.startswith("v_mfma"):
            ops = [o.strip() for o in s[len(op):].split(",")]
            dst, srcb = regs_of(ops[0]), regs_of(ops[2])
            opened_with_nop = idx > 0 and instrs[idx - 1][0].startswith("s_nop") and instrs[idx - 1][1]
            if not opened_with_nop:
                for back in (1, 2):
                    if idx - back < 0:
                        break
                    p, _ = instrs[idx - back]
                    pop = p.split()[0]
                    if pop.startswith("v_") and not pop.startswith("v_mfma") and not pop.startswith("v_cmp"):
                        if regs_of(p.split(",")[0]) & srcb:
                            problems.append(f"{name}: VALU write of an MFMA B operand {back} instruction(s) in front: {p} -> {s}")
            n_mfma += 1
            wait_states = 0
            for r in dst:
                last_write[r] = n_mfma
            continue
        if inside and (op.startswith("v_max3") or op.startswith("v_fma") or op.startswith("ds_write")):
            first, rest = s.split(",", 1) if "," in s else (s, "")
            srcs = regs_of(rest if not op.startswith("ds_write") else s) & ACC
            for r in srcs:
                if r in last_write and n_mfma - last_write[r] < MIN_GAP and wait_states < 12:
                    problems.append(f"{name}: accumulator v{r} read {n_mfma - last_write[r]} MFMAs after its last write "
                                    f"({wait_states} wait states): {s}")
    return problems, n_mfma


def main():
    with tempfile.TemporaryDirectory() as wd:
        path = compile_to_asm(wd)
        text = open(path).read()
        problems = []
        n = 0
        for name, lines in kernels(path):
            p, n_mfma = audit(name, lines)
            problems += p
            n += 1
            print(f"{name[:64]}...: {n_mfma} MFMAs, {len(p)} problems")
        for m in re.finditer(r"\.name:\s+(_ZN5rayen19mfma_pair_ws_kernel\S*)\n\s+\.private_segment_fixed_size:\s+(\d+)(?:.*?\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
            if int(m.group(2)) or int(m.group(3)):
                problems.append(f"{m.group(1)}: scratch {m.group(2)} bytes, {m.group(3)} spilled VGPRs")
        if n == 0:
            problems.append("no instance of mfma_pair_ws_kernel found")
    for p in problems[:50]:
        print("PROBLEM:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
