#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE full PPO iteration of the hot path at BASELINE config[1] (1v1, 4096 envs per GPU, MLP 256x3, bf16 MFMA):
T = 32 gym steps of every env (on-device policy inference + batched arena stepper, tickSkip 8, example
obs/reward/terminal stack, RandomState resets), value predictions, GAE, one epoch of PPO over the B = 8192*32 = 262 144
collected agent-steps in 4 minibatches of 65 536 with one clip+Adam step (and one RCCL gradient all-reduce when N > 1).
Nothing is skipped inside the timed region.  `value` = agent-steps / second over the whole job (the reference's
"steps" unit, ThreadAgent.cpp:158); "PPO iter ms" (the consumption phase: values + GAE + learn) is reported next to it.

The JSON line also carries the dominant kernel's roofline (the env step kernel: algorithmic bytes of SURVEY 8d over
its hipEvent-measured duration) and a CPU baseline of the collection path measured on this host in the same run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def algorithmic_bytes_per_gym_step(n_players, obs_dim):
    # A = 2*(336*Np + 264) + Np*(4*D + 8) + 4
    return 2 * (336 * n_players + 264) + n_players * (4 * obs_dim + 8) + 4


def cpu_baseline(seconds_target=12.0):
    """The reference's CPU collection path on this host's cores (32 envs 1v1 = BASELINE config[0], threads x games like
    ThreadAgent): the real RocketSim/RLGymSim_CPP when the prebuilt oracle/_ref library is present (kind "reference"),
    else the oracle's scalar host build of the stepper (kind "port").  Stepping only, uniform random action tape."""
    import ctypes as C
    cores = os.cpu_count() or 1
    n_envs, team = 32, 1
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so")
    port_so = os.path.join(ROOT, "oracle", "_build", "liboracle_port.so")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # the reference library prints to the C stdout (e.g. "DiscreteAction(): Lookup table built"): keep this process' stdout
    # a single JSON line by parking fd 1 on stderr while the baseline runs
    sys.stdout.flush()
    saved_fd = os.dup(1); os.dup2(2, 1)
    try:
        return _cpu_baseline(seconds_target, cores, n_envs, team, ref_so, port_so)
    finally:
        sys.stdout.flush(); os.dup2(saved_fd, 1); os.close(saved_fd)


def _cpu_baseline(seconds_target, cores, n_envs, team, ref_so, port_so):
    import ctypes as C
    try:
        if os.path.exists(ref_so) and os.path.exists(port_so):
            from simlib import PortSim, RefSim
            port = PortSim(); v, t = port.procedural_mesh()
            ref = RefSim(v, t)
            steps = 200
            sec = ref.lib.ref_bench_collect(team, n_envs, min(cores, n_envs), steps, 8)
            steps = max(200, int(steps * seconds_target / max(sec, 1e-3)))
            sec = ref.lib.ref_bench_collect(team, n_envs, min(cores, n_envs), steps, 8)
            kind = "reference"
        elif os.path.exists(port_so):
            from simlib import PortSim, port_gym_cfg
            port = PortSim(); v, t = port.procedural_mesh(); port.set_mesh(v, t)
            port.lib.port_bench_collect.restype = C.c_double
            cfg = port_gym_cfg()
            steps = 200
            sec = port.lib.port_bench_collect(team, n_envs, min(cores, n_envs), steps, C.byref(cfg))
            steps = max(200, int(steps * seconds_target / max(sec, 1e-3)))
            sec = port.lib.port_bench_collect(team, n_envs, min(cores, n_envs), steps, C.byref(cfg))
            kind = "port"
        else:
            return None
    except Exception as e:  # the baseline must never take the bench down
        return {"value": None, "unit": "agent-steps/s", "cores": cores, "kind": "error", "sample": str(e)[:200]}
    agent_steps = n_envs * 2 * team * steps
    return {"value": agent_steps / sec, "unit": "agent-steps/s", "cores": min(cores, n_envs), "kind": kind,
            "sample": f"{n_envs} envs 1v1 (BASELINE config[0]), {min(cores, n_envs)} threads x {n_envs // min(cores, n_envs)} games, {steps} gym steps/env, "
                      f"tickSkip 8, example obs/reward stack, random actions, procedural mesh, stepping only ({sec:.1f} s)"}


def kernel_source_hash():
    """sha256 over the device sources of the stepper: the key under which profiles/*_pmc.json was recorded (a profile taken from
    other kernel code is not quoted)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rlgymppo_cpp_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")):
            h.update(name.encode()); h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_profile(kernel_prefix):
    """The newest committed rocprofv3 PMC summary for this kernel code (tools/profile_bench.sh writes profiles/rNN*_pmc.json on the GPU
    box): HBM traffic per launch from FETCH_SIZE + WRITE_SIZE (separate passes, KB units; the guide's gfx950 corrections are noted in the
    file) and the SQ figures.  None when the kernel sources changed since."""
    best = None
    pd = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pd)) if os.path.isdir(pd) else []:
        if not name.endswith("_pmc.json"):
            continue
        try:
            j = json.load(open(os.path.join(pd, name)))
        except Exception:
            continue
        if j.get("kernel_source_hash") == kernel_source_hash() and j.get("kernel", "").startswith(kernel_prefix):
            best = j; best["file"] = "profiles/" + name
    return best


def rank_child_env(environ):
    """Environment of the measured process of one rank.  The ranks' bench_main processes meet through a file named after MASTER_PORT and
    a tag common to the launch (rlgpu_comm_init_env); left alone the tag is the parent pid, which here is this script -- one per rank --
    so the tag handed down is THIS script's parent: the launcher (torch.distributed.run's agent), the same for every rank of the node."""
    env = dict(environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if int(env.get("WORLD_SIZE", "1")) > 1:
        env.setdefault("RLGPU_COMM_TAG", "%s_%d" % (env.get("TORCHELASTIC_RUN_ID", "run"), os.getppid()))
    return env


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--team-size", type=int, default=1, help="1 = BASELINE configs[1] (the headline); 2 / 3 with --padded-zero-sum = the shapes of configs[3] / [4]")
    ap.add_argument("--padded-zero-sum", action="store_true", help="DefaultOBSPadded(maxPlayers = team size) + ZeroSumReward around the example stack")
    ap.add_argument("--fp32", action="store_true", help="fp32 MFMA instead of bf16 operands")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trained-warmup", type=int, default=480, help="after the timed region: this many more iterations, then --trained-steps timed ones (the policy has started to play: more contacts per tick); 0 = skip")
    ap.add_argument("--trained-steps", type=int, default=100)
    args = ap.parse_args()

    # The measured process is rlgymppo_cpp_amd/bench_main (C++ on librlgymppo_amd.so / librlgpu.so; HIP runtime + RCCL only).  This
    # script never touches the GPU: it starts that program as a child (one per rank under torchrun: RANK / WORLD_SIZE / LOCAL_RANK /
    # MASTER_PORT are inherited, the ranks meet through rlgpu_comm_init_env), adds the CPU baseline and the committed PMC figures, and
    # prints the line.
    import subprocess
    exe = os.path.join(ROOT, "rlgymppo_cpp_amd", "bench_main")
    if not os.path.exists(exe):
        raise SystemExit("bench.py: rlgymppo_cpp_amd/bench_main is not built (python -c 'import __graft_entry__ as g; g.build()') -- there is no other path")
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch one rank per GPU with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...`")
    cmd = [exe, "--envs", str(args.envs), "--team-size", str(args.team_size), "--horizon", str(args.horizon), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--epochs", str(args.epochs)]
    if args.padded_zero_sum: cmd.append("--padded-zero-sum")
    if args.fp32: cmd.append("--fp32")
    if args.trained_warmup > 0 and args.trained_steps > 0:
        cmd += ["--trained-warmup", str(args.trained_warmup), "--trained-steps", str(args.trained_steps)]
    env = rank_child_env(os.environ)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    if proc.returncode != 0:
        raise SystemExit(f"bench_main failed with exit code {proc.returncode}")
    if rank != 0:
        return
    m = json.loads(proc.stdout.decode().strip().splitlines()[-1])

    n_p = 2 * args.team_size
    A = m["algorithmic_bytes_per_gym_step_per_env"]
    assert A == algorithmic_bytes_per_gym_step(n_p, m["obs_size"])
    # roofline of the dominant kernel, SURVEY 8d verbatim: algorithmic bytes per gym step per env x envs x gym steps per launch / its
    # hipEvent-measured average duration (events on the env's own stream: rlgpu_env_timing_total)
    bytes_per_launch = A * args.envs * m["gym_steps_per_launch"]
    avg_ms = m["env_kernel_ms_total"] / max(1, m["env_launches"])
    gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    kname = (f"k_env_collect<{n_p}> ({args.horizon} x (policy inference + 8 ticks + snapshot/obs/reward/done/auto-reset) in one launch)"
             if m["fused_collect"] else f"k_env_step<{n_p}> (8 ticks + snapshot/obs/reward/done/auto-reset fused)")
    # (the fused launch keeps the arena state in LDS for all its gym steps: counted once per launch instead of once per step the same
    # launch moves `bytes_state_once` -- the 8d figure is the algorithmic upper bound the survey prescribes, this is what must cross HBM)
    state_bytes = 2.0 * (336.0 * n_p + 264.0)
    bytes_state_once = (state_bytes + (A - state_bytes) * m["gym_steps_per_launch"]) * args.envs
    roof = {"kernel": kname, "bound": "hbm", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0, "traffic": None,
            "avg_launch_ms": avg_ms, "launches": m["env_launches"], "algorithmic_bytes_per_launch": bytes_per_launch,
            "formula": "SURVEY 8d: (2*(336*Np+264) + Np*(4*D+8) + 4) B per gym step per env x envs x gym steps per launch",
            "achieved_state_once": bytes_state_once / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0, "bytes_per_launch_state_once": bytes_state_once}
    pmc = pmc_profile("k_env_collect" if m["fused_collect"] else "k_env_step")
    if pmc is not None:
        roof["traffic"] = pmc.get("hbm_bytes_per_launch")
        for k in ("traffic_fetch_bytes", "traffic_write_bytes", "valu_util", "waves_per_simd", "active_lane_fraction", "wait_any_frac", "file", "note"):
            if k in pmc:
                roof["pmc_" + k if not k.startswith("traffic") else k] = pmc[k]
    tflops = m["gemm_flops_total"] / (m["gemm_ms_total"] * 1e-3) / 1e12 if m["gemm_ms_total"] > 0 else 0.0
    peak_tflops = 157.3 if args.fp32 else 2500.0     # dense MFMA peaks, MI355X_MICROARCH.md: fp32 (v_mfma_f32_32x32x2_f32) / bf16
    out = {
        "metric": f"env steps/sec/node + PPO iter ms, {args.team_size}v{args.team_size} {args.envs} envs/GPU",
        "value": m["value"], "unit": "agent-steps/s",
        "n_gpus": m["n_gpus"], "steps": args.steps, "warmup": args.warmup, "ms_per_step": m["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "fp32 stepper + " + ("fp32" if args.fp32 else "bf16") + " MFMA MLP", "data": "synthetic",
        "config": {"workload": (f"BASELINE config[1]: 1v1, {args.envs} envs/GPU, tickSkip 8, DefaultObs(89)+example reward stack, RandomState resets, "
                                if args.team_size == 1 and not args.padded_zero_sum else
                                f"{args.team_size}v{args.team_size}, {args.envs} envs/GPU, tickSkip 8, " + ("DefaultOBSPadded + zero-sum example stack, " if args.padded_zero_sum else "DefaultObs + example stack, ") + "RandomState resets, ")
                               + f"T={args.horizon} steps/iter, B={m['batch']} agent-steps/GPU, minibatch {m['minibatch']}, epochs {args.epochs}, MLP 256x3 policy(90)+critic, procedural arena mesh",
                   "envs_per_gpu": args.envs, "horizon": args.horizon, "batch": m["batch"], "minibatch": m["minibatch"], "epochs": args.epochs,
                   "host": "C++ (rlgymppo_cpp_amd/bench_main on librlgymppo_amd.so; no Python or torch in the measured process)"},
        "ppo_iter_ms": m["ppo_iter_ms"], "gym_steps_per_s": m["value"] / n_p, "physics_ticks_per_s": m["value"] / n_p * 8,
        "collect_ms_per_iter": m["ms_per_step"] - m["ppo_iter_ms"],
        "roofline": roof,
        "mfma": {"kernels": "k_gemm fwd+bwd of policy and critic inside rlgpu_ppo_minibatch (incl. loss kernels)", "achieved_tflops": tflops, "peak_tflops": peak_tflops,
                 "ms_total": m["gemm_ms_total"], "calls": m["gemm_calls"], "frac": tflops / peak_tflops},
        # multi-GPU audit trail (bench_main): RCCL ranks that took part (0 = no communicator), each rank's own ms per iteration, one gradient all-reduce
        "rccl_ranks": m.get("rccl_ranks", 0), "rank_ms_per_step": m.get("rank_ms_per_step", []),
        "allreduce_ms_per_optimizer_step": m.get("allreduce_ms_per_optimizer_step", 0.0), "allreduce_calls": m.get("allreduce_calls", 0),
    }
    if "trained_regime" in m:
        out["trained_regime"] = dict(m["trained_regime"], note="same learner config (1 epoch) continued; at this point play is still close to random. "
                                     "A policy that has learned to chase the ball costs more per tick (DESIGN.md 4.3, profiles/r02l_train_probe.txt: "
                                     "7.9 M agent-steps/s over 1600 two-epoch iterations, Python host)")
    if not args.no_cpu_baseline and world == 1:
        cb = cpu_baseline()
        if cb is not None:
            out["cpu_baseline"] = cb
    print(json.dumps(out))


if __name__ == "__main__":
    main()
