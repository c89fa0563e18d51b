#!/usr/bin/env python
"""Audit of the gfx950 ISA the library compiles to: wide global stores (> 64 bits of data: dwordx3 / dwordx4) whose data VGPRs are
overwritten within a few instructions.

Why: `points16_kernel` wrote its 3-float points with `global_store_dwordx3 vaddr, v[2:4]` and recomputed v2..v4 for the next point six
instructions later (no s_waitcnt in between: the ISA manual only asks for one wait state).  With another kernel's waves on the same CU
(two forwards on two HIP streams) a 16-lane group of the store occasionally carried the NEW x value: the data registers of a wide
store are read late when the memory pipeline is contended.  This script lists every wide store whose data
registers are written again within WINDOW instructions on the fall-through path before an `s_waitcnt vmcnt(..)`.

    tools/check_store_hazard.py [file.s ...]      (default: compiles every translation unit of framedipt_amd/csrc with -save-temps)
"""
import glob
import os
import re
import subprocess
import sys

WINDOW = 16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ST = re.compile(r"^\s*(global|flat|buffer|scratch)_store_dwordx([34])\s+(.*)$")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def dest_regs(ins):
    """VGPRs an instruction writes (first operand of VALU / load / LDS-read / MFMA forms)."""
    ins = ins.strip()
    op = ins.split()[0] if ins else ""
    if not op or op.startswith(("s_", ";", ".")) or "_store_" in op or op.startswith(("ds_write", "ds_bpermute_b32 ")) and False:
        return set()
    if op.startswith(("global_load", "flat_load", "buffer_load", "scratch_load", "ds_read", "ds_bpermute", "v_")):
        if op.startswith(("v_cmp", "v_cmpx")) and not op.endswith("_e64"):
            return set()
        first = ins[len(op):].split(",")[0]
        return regs(first)
    return set()


def scan(path):
    lines = open(path).read().split("\n")
    kern, hits = None, []
    code = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            kern = m.group(1)
        s = l.split(";")[0].rstrip()
        if s.strip() and not s.strip().startswith(".") and not s.endswith(":"):
            code.append((i + 1, kern, s.strip()))
    for k, (ln, kn, ins) in enumerate(code):
        m = ST.match(ins)
        if not m:
            continue
        ops = [t.strip() for t in m.group(3).split(",")]
        data = regs(ops[1]) if m.group(1) != "buffer" else regs(ops[0])
        for j in range(k + 1, min(k + 1 + WINDOW, len(code))):
            nxt = code[j][2]
            if nxt.startswith("s_waitcnt") and "vmcnt" in nxt:
                break
            if nxt.startswith(("s_endpgm", "s_branch", "s_setpc")):
                break
            w = dest_regs(nxt) & data
            if w:
                hits.append((kn, ln, ins, j - k, nxt))
                break
    return hits


def main(argv):
    files = argv
    if not files:
        tmp = "/tmp/fd_hazard"
        os.makedirs(tmp, exist_ok=True)
        for src in sorted(glob.glob(os.path.join(ROOT, "framedipt_amd", "csrc", "*.hip"))):
            base = os.path.basename(src)[:-4]
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-c", src, "-o", f"{tmp}/{base}.o",
                            "-save-temps=obj"], check=True, cwd=os.path.join(ROOT, "framedipt_amd", "csrc"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        files = sorted(glob.glob(f"{tmp}/*-hip-amdgcn-amd-amdhsa-gfx950.s"))
    total = 0
    for f in files:
        for kn, ln, ins, dist, nxt in scan(f):
            total += 1
            print(f"{os.path.basename(f).split('-hip-')[0]}: {kn} line {ln}: {ins}   <- data overwritten {dist} instructions later by: {nxt}")
    print(f"{total} wide stores with an early overwrite of their data registers (window {WINDOW})")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
