#!/usr/bin/env python3
"""ISA lint for one LLVM AMDGPU code-generation defect (found round 4 with rocgdb, DESIGN.md 4.1 "The k_env_collect<4> fault"):

A kernel under full register pressure keeps its SGPR spills in lanes of reserved VGPRs (v_writelane / v_readlane), and spills THOSE VGPRs to
AGPRs or scratch; every such access is bracketed by whole-wave mode --
        s_or_saveexec_b64 sN, -1 ;  <access to the SGPR-spill VGPR> ;  s_mov_b64 exec, sN
-- because the lanes of that VGPR are SGPR values, not per-lane data.  The defect: an UNRELATED per-lane spill copy (v_accvgpr_write aK, vM /
scratch_store of an ordinary VGPR) inserted at the same program point ends up INSIDE the bracket and runs with exec = -1: it then copies the
inactive lanes of vM too, overwriting what another live value had parked in those lanes of aK.

What is NOT the defect, and is recognised as such (round 6): the whole-wave save of a function's PROLOGUE and its mirror in the EPILOGUE.  A function whose
callees use a register in whole-wave mode saves ALL lanes of it on entry (`s_or_saveexec -1 ; scratch_store vK, frame slot ; ... ; s_mov exec`, before the
stack pointer moves and before any branch) and reloads all lanes in front of `s_setpc_b64` -- by design, whatever register it is: the slot belongs to this
activation alone, every lane stored is the lane reloaded, nothing parked in an inactive lane is lost.  (Moving those out of their brackets, as round 5's repair
step did, would have saved the ACTIVE lanes only.)  A prologue store counts only with its mirror reload (same register, same slot) in every epilogue.

usage: wwm_lint.py file.s   (device assembly: hipcc -S --cuda-device-only, or the -save-temps .s)   exit code 1 = suspicious brackets found
"""
import re
import sys

_SPILL = re.compile(r"scratch_store_dword off, ([va]\d+), (s3[23])(?: offset:(\d+))?\s*;.*Folded Spill")
_RELOAD = re.compile(r"scratch_load_dword ([va]\d+), off, (s3[23])(?: offset:(\d+))?\s*;.*Folded Reload")


def _is_code(line):
    return line.startswith("\t") and not line.strip().startswith((";", "."))


def lint(path):
    lines = open(path).read().split("\n")
    func = None
    wwm_regs = set()
    bad = []
    i = 0
    n_brackets = 0
    func_start = 0
    func_end = 0
    candidates = []   # (func, line, ins, ctx, kind, reg, slot) -- prologue / epilogue saves waiting for their mirror
    while i < len(lines):
        l = lines[i]
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1); wwm_regs = set(); func_start = i
            # the SGPR-spill VGPRs of this function: every VGPR that is the destination of a v_writelane / source of a v_readlane
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                mm = re.match(r"\s+v_writelane_b32 (v\d+),", lines[j]) or re.match(r"\s+v_readlane_b32 s\d+, (v\d+),", lines[j])
                if mm:
                    wwm_regs.add(mm.group(1))
                j += 1
            func_end = j
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
            # where the bracket sits: in the entry block before the stack pointer moves (prologue), or straight in front of the return (epilogue)
            pre = [lines[k].strip() for k in range(func_start + 1, i) if _is_code(lines[k])]
            in_prologue = not any(p.startswith(("s_cbranch", "s_branch", "s_swappc", "s_setpc", "s_addk_i32 s32", "s_add_i32 s32")) for p in pre) and \
                not any(re.match(r"^\.LBB", lines[k]) for k in range(func_start + 1, i))
            in_epilogue = False
            for k in range(j + 1, min(j + 24, func_end)):
                t = lines[k].strip()
                if re.match(r"^\.LBB", lines[k]) or t.startswith(("s_cbranch", "s_branch", "s_swappc")):
                    break
                if t.startswith("s_setpc_b64"):
                    in_epilogue = True; break
            for (k, ins) in inside:
                if ins.startswith(("s_waitcnt", "s_nop")):
                    continue
                regs = set(re.findall(r"\bv\d+\b", ins))
                if not regs or not regs <= wwm_regs:
                    ms, ml = _SPILL.match(ins), _RELOAD.match(ins)
                    if in_prologue and ms:
                        candidates.append((func, k + 1, ins, [x for _, x in inside], "save", ms.group(1), ms.group(3) or "0"))
                    elif in_epilogue and ml:
                        candidates.append((func, k + 1, ins, [x for _, x in inside], "restore", ml.group(1), ml.group(3) or "0"))
                    else:
                        bad.append((func, k + 1, ins, [x for _, x in inside]))
            i = j
        i += 1
    # a prologue save is what it looks like only with its mirror in an epilogue, and the other way round
    saves = {(c[0], c[5], c[6]) for c in candidates if c[4] == "save"}
    restores = {(c[0], c[5], c[6]) for c in candidates if c[4] == "restore"}
    for c in candidates:
        key = (c[0], c[5], c[6])
        if not (key in saves and key in restores):
            bad.append(c[:4])
    return n_brackets, bad


if __name__ == "__main__":
    n, bad = lint(sys.argv[1])
    print(f"{n} whole-wave brackets, {len(bad)} instruction(s) inside one that do not touch an SGPR-spill VGPR")
    for func, line, ins, ctx in bad:
        print(f"  {func[:80]} line {line}: {ins}    [bracket: {' ; '.join(ctx)}]")
    sys.exit(1 if bad else 0)
