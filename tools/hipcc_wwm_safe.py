#!/usr/bin/env python3
"""hipcc for one translation unit with the device assembly checked -- and, when needed, repaired -- for the whole-wave-bracket defect
tools/wwm_lint.py describes (LLVM AMDGPU, ROCm 7.2 clang 22: an ordinary per-lane spill copy placed inside the `s_or_saveexec_b64 sN, -1 ...
s_mov_b64 exec, sN` bracket that guards an access to an SGPR-spill VGPR, so that it runs for ALL lanes and destroys what inactive lanes had
parked in the destination; found with rocgdb in k_env_collect<4>).

    tools/hipcc_wwm_safe.py [--log FILE] <hipcc arguments for ONE -c compile>

Runs hipcc's own four steps (hipcc -###: device cc1, lld, clang-offload-bundler, host cc1) with the device step split in two -- cc1 -S, then
the assembler -- and between them moves every instruction the lint flags to just in front of its bracket, where it runs under the exec mask
of its own program point (it depends on nothing inside the bracket: checked).  The object is otherwise what plain hipcc makes.
"""
import os
import re
import shlex
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import wwm_lint  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def patch(path, log):
    n, bad = wwm_lint.lint(path)
    if not bad:
        log.write(f"wwm: {n} whole-wave brackets, clean\n")
        return 0
    lines = open(path).read().split("\n")
    flagged = sorted({ln - 1 for _, ln, _, _ in bad})

    def regs_of(text):
        r = set(re.findall(r"\b[vas]\d+\b", text))
        for m in re.finditer(r"\b([vas])\[(\d+):(\d+)\]", text):
            r |= {f"{m.group(1)}{k}" for k in range(int(m.group(2)), int(m.group(3)) + 1)}
        return r
    # group the flagged lines by the bracket they sit in (its opening s_or_saveexec line)
    groups = {}
    for idx in flagged:
        j = idx
        while not re.match(r"\s+s_or_saveexec_b64 s\[\d+:\d+\], -1", lines[j]):
            j -= 1
            if idx - j > 40:
                raise SystemExit(f"wwm: no bracket start above line {idx + 1}")
        groups.setdefault(j, []).append(idx)
    moved = 0
    for j in sorted(groups, reverse=True):          # bottom-up: indices above stay valid
        idxs = groups[j]
        keep = [k for k in range(j, max(idxs) + 1) if k not in idxs]
        drain = False
        for idx in idxs:
            # independence: no register of a moved instruction is touched by the bracket's own instructions it jumps over ...
            for k in keep:
                if k < idx and regs_of(lines[idx]) & regs_of(lines[k]):
                    raise SystemExit(f"wwm: cannot move `{lines[idx].strip()}` (line {idx + 1}) in front of its bracket: shares {sorted(regs_of(lines[idx]) & regs_of(lines[k]))} with `{lines[k].strip()}`")
                # ... and it jumps over nothing that orders it: a wait (the moved instruction may read a register whose load that s_waitcnt is there
                # for: in front of it the value would be stale), or a writer of exec / vcc other than the bracket's own first instruction
                if j < k < idx:
                    ins = lines[k].strip()
                    if re.search(r"\b(exec|vcc)(_lo|_hi)?\b", ins.split(";")[0].split(",")[0]):
                        raise SystemExit(f"wwm: cannot move `{lines[idx].strip()}` (line {idx + 1}) in front of its bracket: `{ins}` lies between (an exec / vcc writer)")
                    if ins.startswith("s_waitcnt"):
                        # The moved instruction may read a register whose load that wait is there for.  The same counter value would promise LESS at the
                        # earlier position (memory operations issued in between count towards it), so the instruction takes a full drain with it: always
                        # sufficient, and these are spill copies on rare paths.
                        drain = True
        out = (["\ts_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)\t; (a wait lay between the bracket's start and a moved instruction: tools/hipcc_wwm_safe.py)"] if drain else []) + \
              [lines[idx] + "\t; moved in front of the whole-wave bracket (tools/hipcc_wwm_safe.py)" for idx in idxs]
        if drain: log.write(f"wwm: a full s_waitcnt goes with the instruction(s) moved out of the bracket at line {j + 1}\n")
        for idx in idxs:
            log.write(f"wwm: moved `{lines[idx].strip()}` out of the bracket at line {j + 1} ({[b for b in bad if b[1] - 1 == idx][0][0][:70]})\n")
        for idx in reversed(idxs):
            del lines[idx]
        lines[j:j] = out + ["\ts_nop 1"]
        moved += len(idxs)
    open(path, "w").write("\n".join(lines))
    n2, bad2 = wwm_lint.lint(path)
    if bad2:
        raise SystemExit(f"wwm: {len(bad2)} flagged instruction(s) left after patching")
    return moved


def main():
    args = sys.argv[1:]
    log = sys.stderr
    if args and args[0] == "--log":
        log = open(args[1], "w"); args = args[2:]
    out = subprocess.run([HIPCC] + args + ["-###"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    cmds = [shlex.split(l) for l in out.stdout.split("\n") if l.startswith(' "')]
    if len(cmds) != 4 or "-fcuda-is-device" not in cmds[0] or "-emit-obj" not in cmds[0]:
        # not the four steps this script knows how to split (another ROCm, extra flags): build with plain hipcc, and make the lint a HARD check on
        # the device assembly of the same command -- an object with the defect in it is never produced silently
        log.write("wwm: unexpected hipcc pipeline (%d steps): plain hipcc + lint of the device assembly\n" % len(cmds))
        r = subprocess.run([HIPCC] + args)
        if r.returncode: sys.exit(r.returncode)
        if "-o" in args:
            asm = args[args.index("-o") + 1] + ".wwm_check.s"
            sargs = [a for a in args if a != "-c"]; sargs[sargs.index("-o") + 1] = asm
            r = subprocess.run([HIPCC] + sargs + ["-S", "--cuda-device-only"])
            if r.returncode: sys.exit(r.returncode)
            n, bad = wwm_lint.lint(asm)
            os.remove(asm)
            if bad:
                raise SystemExit(f"wwm: {len(bad)} instruction(s) inside a whole-wave bracket that do not belong there, and no way to repair them with this toolchain: " + "; ".join(b[2] for b in bad[:4]))
            log.write(f"wwm: {n} whole-wave brackets, clean\n")
        return
    dev, lld, bundle, host = cmds
    dev_obj = dev[dev.index("-o") + 1]
    dev_asm = dev_obj[:-2] + ".s"
    dev_s = list(dev); dev_s[dev_s.index("-emit-obj")] = "-S"; dev_s[dev_s.index("-o") + 1] = dev_asm
    tmp = [dev_asm, dev_obj, lld[lld.index("-o") + 1], [a for a in bundle if a.startswith("-output=")][0][8:]]
    try:
        r = subprocess.run(dev_s)
        if r.returncode: sys.exit(r.returncode)
        moved = patch(dev_asm, log)
        clang = dev[0]
        mcpu = dev[dev.index("-target-cpu") + 1] if "-target-cpu" in dev else "gfx950"   # (the --offload-arch the caller asked for)
        r = subprocess.run([clang, "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=" + mcpu, "-c", dev_asm, "-o", dev_obj])
        if r.returncode: sys.exit(r.returncode)
        for c in (lld, bundle, host):
            r = subprocess.run(c)
            if r.returncode: sys.exit(r.returncode)
        if moved:
            log.write(f"wwm: object built from the repaired assembly ({moved} instruction(s) moved)\n")
    finally:
        for t in tmp:
            try: os.remove(t)
            except OSError: pass
        if log is not sys.stderr: log.close()


if __name__ == "__main__":
    main()
