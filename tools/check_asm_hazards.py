#!/usr/bin/env python3
"""The packed-add helpers of common.h are inline asm, which the compiler's hazard recogniser treats as opaque.
This scans the generated ISA of conv.hip for the two MFMA hazards an opaque VALU write could hit:
  RAW  VALU write -> v_mfma read of that VGPR within 2 wait states (covered by the s_nop inside the asm)
  WAR  v_mfma (16x16) reads SrcC -> VALU write of that VGPR within 7 wait states
  RAW2 v_mfma writes vDst -> asm read of it (asm inputs must never be MFMA results)
Usage: tools/check_asm_hazards.py [conv.s]   (exit code 1 on a finding)"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/curla_conv_hazard.s"
    if len(sys.argv) <= 1:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S",
                               "--cuda-device-only", "-o", path, os.path.join(ROOT, "curla_amd/csrc/conv.hip")],
                              stderr=subprocess.DEVNULL)
    insts = []  # (kind, dst, srcs, srcC, line_no, text)
    in_asm = False
    for n, line in enumerate(open(path), 1):
        t = line.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        ops = [o.strip() for o in t[len(op):].split(",")]
        ops = [o.split()[0] for o in ops if o]
        if op.startswith("v_mfma"):
            insts.append(("mfma", regs(ops[0]), regs(ops[1]) | regs(ops[2]), regs(ops[3]) if len(ops) > 3 else set(), n, t))
        elif in_asm and op.startswith("v_"):
            insts.append(("asm", regs(ops[0]), set().union(*[regs(o) for o in ops[1:]]), set(), n, t))
        elif op == "s_nop":
            for _ in range(int(ops[0]) + 1):
                insts.append(("nop", set(), set(), set(), n, t))
        else:
            insts.append(("other", set(), set(), set(), n, t))
    bad = 0
    nasm = 0
    for i, (kind, dst, srcs, _, n, t) in enumerate(insts):
        if kind != "asm":
            continue
        nasm += 1
        for k in range(1, 8):  # look back: MFMA SrcC read -> this write (WAR, 7 wait states)
            if i - k < 0:
                break
            pk, pdst, psrcs, pc, pn, pt = insts[i - k]
            if pk == "mfma" and (pc & dst):
                print(f"WAR  line {n}: {t}\n     overwrites SrcC of line {pn} ({k - 1} wait states): {pt}")
                bad += 1
        for k in range(1, 20):  # look back: MFMA result -> asm read
            if i - k < 0:
                break
            pk, pdst, psrcs, pc, pn, pt = insts[i - k]
            if pk == "mfma" and (pdst & srcs):
                print(f"RAW2 line {n}: {t}\n     reads the result of line {pn}: {pt}")
                bad += 1
        for k in range(1, 3):  # look ahead: this write -> MFMA read (RAW, 2 wait states)
            if i + k >= len(insts):
                break
            nk, ndst, nsrcs, nc, nn, nt = insts[i + k]
            if nk == "mfma" and (dst & (nsrcs | nc)):
                print(f"RAW  line {n}: {t}\n     read {k - 1} wait states later by line {nn}: {nt}")
                bad += 1
    print(f"{nasm} inline-asm VALU ops checked, {bad} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
