#!/usr/bin/env python3
"""Per-function register / scratch report of the stepper's device assembly (every function, not only the kernels -Rpass-analysis prints).

    tools/asm_report.py <file.s> [name filter]            report of an assembly file (hipcc -S --cuda-device-only)
    tools/asm_report.py --build <tag> [hipcc -D flags]    compiles csrc/rlgpu_env.hip to /tmp/v/<tag>.s first

Per function: code bytes, VGPRs, AGPRs, the frame (private_seg_size: the function's OWN frame; a kernel's ScratchSize is the maximum over its call
graph), scratch loads / stores in its body, SGPR spill copies (v_writelane / v_readlane), whole-wave brackets (s_or_saveexec -1) and what
tools/wwm_lint.py flags in them.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import wwm_lint  # noqa: E402


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"rlg::", "", n)
    m = re.match(r"(?:void |int |bool |float )?([\w:]+(?:<[^()]*?>)?)\(", n)
    return (m.group(1) if m else n)[:60]


def report(path, flt=None):
    funcs = []
    cur = None
    for ln in open(path, errors="replace"):
        m = re.match(r"\s+\.type\s+(\S+),@function", ln)
        if m:
            cur = {"name": m.group(1), "sld": 0, "sst": 0, "wl": 0, "rl": 0, "wwm": 0, "calls": 0, "kernel": False}
            funcs.append(cur); continue
        if cur is None:
            continue
        s = ln.strip()
        if s.startswith("scratch_load"): cur["sld"] += 1
        elif s.startswith("scratch_store"): cur["sst"] += 1
        elif s.startswith("v_writelane_b32"): cur["wl"] += 1
        elif s.startswith("v_readlane_b32"): cur["rl"] += 1
        elif s.startswith("s_or_saveexec_b64") and s.endswith("-1"): cur["wwm"] += 1
        elif s.startswith("s_swappc_b64"): cur["calls"] += 1
        elif s.startswith("; Kernel info"): cur["kernel"] = True
        else:
            for key, pat in (("code", r"; codeLenInByte = (\d+)"), ("vgpr", r"; NumVgprs: (\d+)"), ("agpr", r"; NumAgprs: (\d+)"), ("scratch", r"; ScratchSize: (\d+)"),
                             ("sgpr_spill", r"; SGPRSpill.*?(\d+)"), ("vgpr_spill", r"; VGPRSpill.*?(\d+)"), ("sgprs", r"; TotalNumSgprs: (\d+)")):
                mm = re.match(pat, s)
                if mm and key not in cur: cur[key] = int(mm.group(1))
    n_br, bad = wwm_lint.lint(path)
    per = {}
    for b in bad:
        per[b[0]] = per.get(b[0], 0) + 1
    dm = demangle([f["name"] for f in funcs])
    print(f"{'function':62s} {'code':>7s} {'vgpr':>4s} {'agpr':>4s} {'frame':>6s} {'s_ld':>5s} {'s_st':>5s} {'wlane':>5s} {'rlane':>5s} {'wwm':>4s} {'calls':>5s} {'flag':>4s}")
    for f in funcs:
        name = short(dm.get(f["name"], f["name"]))
        if flt and flt not in name: continue
        flagged = sum(v for k, v in per.items() if k.startswith(f["name"][:70]))
        print(f"{('K ' if f['kernel'] else '  ') + name:62s} {f.get('code', 0):7d} {f.get('vgpr', 0):4d} {f.get('agpr', 0):4d} {f.get('scratch', 0):6d} {f['sld']:5d} {f['sst']:5d} {f['wl']:5d} {f['rl']:5d} {f['wwm']:4d} {f['calls']:5d} {flagged:4d}"
              + (f"  sgpr_spill {f['sgpr_spill']} vgpr_spill {f['vgpr_spill']}" if f.get("sgpr_spill") is not None and f["kernel"] else ""))
    print(f"whole-wave brackets: {n_br}, flagged instructions: {len(bad)}")


def main():
    a = sys.argv[1:]
    if a and a[0] == "--build":
        tag = a[1]; flags = a[2:]
        os.makedirs("/tmp/v", exist_ok=True)
        out = f"/tmp/v/{tag}.s"
        src = os.path.join(HERE, "..", "rlgymppo_cpp_amd", "csrc", "rlgpu_env.hip")
        cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-value", "-ffp-contract=off", "-S", "--cuda-device-only",
               "-Rpass-analysis=kernel-resource-usage"] + flags + [src, "-o", out]
        r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        open(f"/tmp/v/{tag}.resource.log", "w").write(r.stderr)
        if r.returncode:
            sys.stderr.write(r.stderr[-4000:]); sys.exit(r.returncode)
        # the kernels' spill counts come from the remarks
        ks = re.findall(r"Function Name: (\S+).*?ScratchSize \[bytes/lane\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", r.stderr, re.S)
        dm = demangle([k[0] for k in ks])
        for k in ks:
            print(f"{short(dm[k[0]]):40s} ScratchSize {k[1]:>5s}  SGPR spills {k[2]:>4s}  VGPR spills {k[3]:>4s}")
        report(out, None)
    else:
        report(a[0], a[1] if len(a) > 1 else None)


if __name__ == "__main__":
    main()
