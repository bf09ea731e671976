"""profiles/<tag>_pmc.json from the pass summaries tools/profile_bench.sh wrote: per-launch HBM traffic and the SQ occupancy / issue
figures of the dominant kernel, keyed by the hash of the kernel sources so that bench.py only quotes them for the code they measure."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash

out_dir, tag = sys.argv[1], sys.argv[2]


def counters(name):
    """{kernel prefix: {counter: (sum, dispatches)}} from a read_prof.py summary."""
    res = {}
    path = os.path.join(out_dir, name + ".summary.txt")
    if not os.path.exists(path):
        return res
    sec = False
    for line in open(path):
        if line.startswith("== counters"):
            sec = True; continue
        if not sec:
            continue
        m = re.match(r"(.{60}) (\S+)\s+(\d+)\s+(\d+)\s*$", line.rstrip("\n"))
        if m:
            res.setdefault(m.group(1).strip(), {})[m.group(2)] = (float(m.group(3)), int(m.group(4)))
    return res


def kernel_avg_us(prefix):
    path = os.path.join(out_dir, "kt.summary.txt")
    for line in open(path) if os.path.exists(path) else []:
        if prefix in line and not line.startswith("=="):
            f = line.split()
            return float(f[-2]), int(f[-4])
    return None, None


allc = {}
for p in ("fetch", "write", "sq1", "sq2"):
    for k, v in counters(p).items():
        allc.setdefault(k, {}).update(v)
dom = None
for k in allc:
    if "k_env_collect" in k or (dom is None and "k_env_step" in k):
        dom = k
if dom is None:
    print(json.dumps({"error": "no env kernel in the profile", "kernel_source_hash": kernel_source_hash()})); sys.exit(0)
c = allc[dom]
g = lambda n: c[n][0] / max(1, c[n][1]) if n in c else None     # per launch
kname = "k_env_collect" if "k_env_collect" in dom else "k_env_step"
avg_us, calls = kernel_avg_us(kname)
j = {"tag": tag, "kernel": kname, "kernel_full_name": dom, "kernel_source_hash": kernel_source_hash(), "kernel_trace_avg_us": avg_us, "kernel_trace_calls": calls,
     "raw_per_launch": {n: g(n) for n in sorted(c)}}
fetch_kb, write_kb = g("FETCH_SIZE"), g("WRITE_SIZE")
if fetch_kb is not None and write_kb is not None:
    # rocprofv3 reports both in KB.  MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-byte requests of WIDE coalesced streaming
    # reads at 64 B (reads half) and is uncalibrated for other access widths; WRITE_SIZE is exact for 16-B-per-lane stores and dword
    # atomics.  This kernel's global accesses are 4-byte words (state rows, observation rows, scratch) -- outside the calibrated patterns,
    # so the raw KB figures are quoted and the doubled-fetch figure is given as the upper bracket.
    j["traffic_fetch_bytes"] = fetch_kb * 1024.0; j["traffic_write_bytes"] = write_kb * 1024.0
    j["hbm_bytes_per_launch"] = (fetch_kb + write_kb) * 1024.0
    j["hbm_bytes_per_launch_fetch_doubled"] = (2 * fetch_kb + write_kb) * 1024.0
    j["note"] = "FETCH_SIZE + WRITE_SIZE (KB -> bytes), separate passes; 4-byte access pattern = uncalibrated per MI355X_MICROARCH.md (HBM), fetch-doubled bracket alongside"
wc, busy, waves = g("SQ_WAVE_CYCLES"), g("SQ_BUSY_CYCLES"), g("SQ_WAVES")
if wc:
    # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (guide: rocprofv3 PMC slots / cycle constants)
    if g("SQ_ACTIVE_INST_VALU") is not None: j["valu_util"] = g("SQ_ACTIVE_INST_VALU") / wc          # share of a resident wave's time spent issuing VALU
    if g("SQ_WAIT_ANY") is not None: j["wait_any_frac"] = g("SQ_WAIT_ANY") / wc
    if g("SQ_WAIT_INST_ANY") is not None: j["wait_inst_any_frac"] = g("SQ_WAIT_INST_ANY") / wc
    if g("SQ_ACTIVE_INST_ANY") is not None: j["active_inst_any_frac"] = g("SQ_ACTIVE_INST_ANY") / wc
if g("SQ_THREAD_CYCLES_VALU") is not None and g("SQ_ACTIVE_INST_VALU"):
    j["active_lane_fraction"] = g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0)          # active lanes per issued VALU instruction / 64
if waves and avg_us:
    # resident waves per SIMD averaged over the launch: wave-quad-cycles x 4 / (launch cycles x 1024 SIMDs)
    clk = 2.4e3   # MHz nominal; GRBM_GUI_ACTIVE / 8 / time when collected
    if g("GRBM_GUI_ACTIVE") is not None: clk = g("GRBM_GUI_ACTIVE") / 8.0 / avg_us
    j["effective_clock_mhz"] = clk
    if wc: j["waves_per_simd"] = wc * 4.0 / (avg_us * clk * 1024.0)
    j["waves_launched"] = waves
print(json.dumps(j, indent=1))
