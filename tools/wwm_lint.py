#!/usr/bin/env python3
"""ISA lint for one LLVM AMDGPU code-generation defect (found round 4 with rocgdb, DESIGN.md 4.1 "The k_env_collect<4> fault"):

A kernel under full register pressure keeps its SGPR spills in lanes of reserved VGPRs (v_writelane / v_readlane), and spills THOSE VGPRs to
AGPRs or scratch; every such access is bracketed by whole-wave mode --
        s_or_saveexec_b64 sN, -1 ;  <access to the SGPR-spill VGPR> ;  s_mov_b64 exec, sN
-- because the lanes of that VGPR are SGPR values, not per-lane data.  The defect: an UNRELATED per-lane spill copy (v_accvgpr_write aK, vM /
scratch_store of an ordinary VGPR) inserted at the same program point ends up INSIDE the bracket and runs with exec = -1: it then copies the
inactive lanes of vM too, overwriting what another live value had parked in those lanes of aK.

usage: wwm_lint.py file.s   (device assembly: hipcc -S --cuda-device-only)   exit code 1 = suspicious brackets found
"""
import re
import sys


def lint(path):
    lines = open(path).read().split("\n")
    func = None
    wwm_regs = set()
    bad = []
    i = 0
    n_brackets = 0
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1); wwm_regs = set()
            # the SGPR-spill VGPRs of this function: every VGPR that is the destination of a v_writelane / source of a v_readlane
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                mm = re.match(r"\s+v_writelane_b32 (v\d+),", lines[j]) or re.match(r"\s+v_readlane_b32 s\d+, (v\d+),", lines[j])
                if mm:
                    wwm_regs.add(mm.group(1))
                j += 1
        m = re.match(r"\s+s_or_saveexec_b64 (s\[\d+:\d+\]), -1", l)
        if m and func:
            save = m.group(1)
            j = i + 1
            inside = []
            while j < len(lines) and not re.match(r"\s+s_mov_b64 exec, " + re.escape(save), lines[j]):
                if lines[j].startswith("\t") and not lines[j].strip().startswith(";"):
                    inside.append((j, lines[j].strip()))
                j += 1
                if j - i > 40:
                    break
            n_brackets += 1
            for (k, ins) in inside:
                if ins.startswith(("s_waitcnt", "s_nop")):
                    continue
                regs = set(re.findall(r"\bv\d+\b", ins))
                if not regs or not regs <= wwm_regs:
                    bad.append((func, k + 1, ins, [x for _, x in inside]))
            i = j
        i += 1
    return n_brackets, bad


if __name__ == "__main__":
    n, bad = lint(sys.argv[1])
    print(f"{n} whole-wave brackets, {len(bad)} instruction(s) inside one that do not touch an SGPR-spill VGPR")
    for func, line, ins, ctx in bad:
        print(f"  {func[:80]} line {line}: {ins}    [bracket: {' ; '.join(ctx)}]")
    sys.exit(1 if bad else 0)
