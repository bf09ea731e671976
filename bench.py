#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE full PPO iteration of the hot path at BASELINE config[1] (1v1, 4096 envs per GPU, MLP 256x3, bf16 MFMA):
timestepsPerIteration = B = 8192*32 = 262 144 agent-steps gathered by the 4096 envs (on-device policy inference + batched arena stepper,
tickSkip 8, example obs/reward/terminal stack, RandomState resets) the way the reference's agent threads gather them -- every game at its
own pace until the batch has them together (ThreadAgentManager.cpp:16-82; `collection: free-running`, on average 33 gym steps per env,
B .. B + one step of every game per iteration; `--lockstep` makes every env take exactly T = 32 steps, reported as the `lockstep_collection`
leg) --, value predictions, GAE, one epoch of PPO over B rows in 4 minibatches of 65 536 with one clip+Adam step (and one RCCL gradient
all-reduce when N > 1).  Nothing is skipped inside the timed region.  `value` = agent-steps REALLY gathered / second over the whole job (the
reference's "steps" unit, ThreadAgent.cpp:158); "PPO iter ms" (the consumption phase: values + GAE + learn) is reported next to it.

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


def cpu_baseline(seconds_target=12.0, mesh_dir=None):
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
        return _cpu_baseline(seconds_target, cores, n_envs, team, ref_so, port_so, mesh_dir)
    finally:
        sys.stdout.flush(); os.dup2(saved_fd, 1); os.close(saved_fd)


def _cpu_baseline(seconds_target, cores, n_envs, team, ref_so, port_so, mesh_dir=None):
    import ctypes as C
    try:
        if mesh_dir is not None and not os.path.exists(ref_so):
            return None          # a mesh of several files is the reference's own loader's business; the port baseline is quoted on the one-object mesh only
        if os.path.exists(ref_so) and os.path.exists(port_so):
            from simlib import PortSim, RefSim
            port = PortSim(); v, t = port.procedural_mesh()
            ref = RefSim(v, t, mesh_dir=mesh_dir)
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
                      f"tickSkip 8, example obs/reward stack, random actions, {'the tessellated mesh files' if mesh_dir else 'procedural mesh'}, stepping only ({sec:.1f} s)"}


def kernel_source_hash():
    """sha256 over the device sources of the stepper and the learner kernels: the key under which profiles/*_pmc.json was recorded (a
    profile taken from other kernel code is not quoted).  The RCCL glue (rlgpu_comm.hip) launches no kernel of its own and is left out."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "rlgymppo_cpp_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip")) and name != "rlgpu_comm.hip":
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
    ap.add_argument("--fp16", action="store_true", help="fp16 operands + dynamic loss scale in the minibatch kernels (BASELINE configs[4]: 'fp16 autocast MFMA')")
    ap.add_argument("--overlap", action="store_true", help="LearnerConfig::collectionDuringLearn: the epochs run beside the next collection (BASELINE configs[4])")
    ap.add_argument("--lockstep", action="store_true", help="every env takes exactly --horizon steps per iteration (LearnerConfig::lockstepCollection) instead of the reference's free-running agents")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--trained-warmup", type=int, default=480, help="after the timed region: this many more iterations, then --trained-steps timed ones (the policy has started to play: more contacts per tick); 0 = skip")
    ap.add_argument("--trained-steps", type=int, default=100)
    ap.add_argument("--learned-warmup", type=int, default=1000, help="then: this many more iterations of LEARNING with --learned-epochs PPO epochs each (the policy then chases and hits the ball), "
                    "and --learned-steps timed iterations with the headline's settings: `trained_regime_learned`; 0 = skip")
    ap.add_argument("--learned-epochs", type=int, default=2)
    ap.add_argument("--learned-steps", type=int, default=100)
    ap.add_argument("--mesh", choices=["procedural", "tessellated", "both"], default="both",
                    help="procedural: the 180-triangle arena (the headline); tessellated: the same arena at ~9 k triangles in 16 .cmf files (what the game's own soccar set looks "
                         "like to the stepper), given to the HIP path and to the reference baseline through their directory loaders; both: headline + a `mesh_tessellated` leg")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the short legs of BASELINE configs[3] (2v2, 8192 envs, padded obs + zero-sum) and configs[4] "
                    "(3v3, 16384 envs, collect-during-learn + fp16 operands) that follow the headline (`configs` in the line; 1 GPU only)")
    ap.add_argument("--no-user-reward-leg", action="store_true", help="skip the leg that runs the headline with a user RewardFunction subclass (host reward, deferred: `user_reward` in the line)")
    ap.add_argument("--allow-test-transport", action="store_true", help="tests only: print a line for N > 1 although the gradients went over the host-staged "
                    "shared-memory transport (RLGPU_COMM_TRANSPORT=shm, ranks sharing one GPU) instead of RCCL; such a line carries \"transport\": \"shm\"")
    ap.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)   # internal: print the CPU baseline for the mesh directory given, as JSON
    args = ap.parse_args()

    # The measured process is rlgymppo_cpp_amd/bench_main (C++ on librlgymppo_amd.so / librlgpu.so; HIP runtime + RCCL only).  This
    # script never touches the GPU: it starts that program as a child (one per rank under torchrun: RANK / WORLD_SIZE / LOCAL_RANK /
    # MASTER_PORT are inherited, the ranks meet through rlgpu_comm_init_env), adds the CPU baseline and the committed PMC figures, and
    # prints the line.
    import subprocess
    if args.cpu_baseline_child is not None:      # a process initialises the reference once: the second mesh gets a process of its own
        print(json.dumps(cpu_baseline(8.0, mesh_dir=args.cpu_baseline_child)))
        return
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
    if args.fp16: cmd.append("--fp16")
    if args.overlap: cmd.append("--overlap")
    if args.lockstep: cmd.append("--lockstep")
    if args.trained_warmup > 0 and args.trained_steps > 0:
        cmd += ["--trained-warmup", str(args.trained_warmup), "--trained-steps", str(args.trained_steps)]
    if args.learned_warmup > 0 and args.learned_steps > 0 and world == 1:
        cmd += ["--learned-warmup", str(args.learned_warmup), "--learned-epochs", str(args.learned_epochs), "--learned-steps", str(args.learned_steps)]
    env = rank_child_env(os.environ)
    if args.mesh == "tessellated":
        mesh_dir, mesh_info = make_tessellated_mesh_dir()
        cmd += ["--mesh-dir", mesh_dir]
    ckpt_dir = None
    if world == 1 and args.mesh == "both" and args.learned_warmup > 0 and args.learned_steps > 0:
        # the learned policy is kept (Learner::Save) so that the tessellated mesh can be measured under it too: what a real user runs
        import tempfile
        ckpt_dir = tempfile.mkdtemp(prefix="rlgpu_ckpt_")
        cmd += ["--save-checkpoint", ckpt_dir]
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, cwd=ROOT)
    if proc.returncode != 0:
        raise SystemExit(f"bench_main failed with exit code {proc.returncode}")
    if rank != 0:
        return
    m = json.loads(proc.stdout.decode().strip().splitlines()[-1])
    # a multi-GPU line is a scaling result only when the gradients went over RCCL: the shared-memory test transport (ranks on ONE device) is refused
    transport = m.get("transport", "none")
    if m["n_gpus"] > 1 and transport != "rccl" and not args.allow_test_transport:
        raise SystemExit(f"bench.py: {m['n_gpus']} ranks exchanged gradients over the '{transport}' transport, not RCCL: not a benchmark line (tests pass --allow-test-transport)")

    n_p = 2 * args.team_size
    A = m["algorithmic_bytes_per_gym_step_per_env"]
    assert A == algorithmic_bytes_per_gym_step(n_p, m["obs_size"])
    # roofline of the dominant kernel, SURVEY 8d verbatim: algorithmic bytes per gym step per env x envs x gym steps per launch / its
    # hipEvent-measured average duration (events on the env's own stream: rlgpu_env_timing_total)
    bytes_per_launch = A * args.envs * m["gym_steps_per_launch"]
    avg_ms = m["env_kernel_ms_total"] / max(1, m["env_launches"])
    gbps = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    kname = (f"k_env_collect<{n_p}> ({m['gym_steps_per_launch']:.1f} x (policy inference + 8 ticks + snapshot/obs/reward/done/auto-reset) in one launch)"
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
        "dtype": "fp32 stepper + " + ("fp32" if args.fp32 else ("fp16 (dynamic loss scale)" if args.fp16 else "bf16")) + " MFMA MLP", "data": "synthetic",
        "config": {"workload": (f"BASELINE config[1]: 1v1, {args.envs} envs/GPU, tickSkip 8, DefaultObs(89)+example reward stack, RandomState resets, "
                                if args.team_size == 1 and not args.padded_zero_sum else
                                f"{args.team_size}v{args.team_size}, {args.envs} envs/GPU, tickSkip 8, " + ("DefaultOBSPadded + zero-sum example stack, " if args.padded_zero_sum else "DefaultObs + example stack, ") + "RandomState resets, ")
                               + f"T={args.horizon} steps/iter, B={m['batch']} agent-steps/GPU, minibatch {m['minibatch']}, epochs {args.epochs}, MLP 256x3 policy(90)+critic, procedural arena mesh",
                   "collection": m.get("collection", "lockstep") + (" (every game at its own pace until the batch holds B agent-steps: ThreadAgentManager.cpp:16-82; %.2f gym steps per env and iteration)" % m["gym_steps_per_launch"]
                                                                     if m.get("collection") == "free-running" else ""),
                   "agent_steps_per_iter": m["agent_steps"] / max(1, args.steps) / max(1, m["n_gpus"]),
                   "envs_per_gpu": args.envs, "horizon": args.horizon, "batch": m["batch"], "minibatch": m["minibatch"], "epochs": args.epochs,
                   "host": "C++ (rlgymppo_cpp_amd/bench_main on librlgymppo_amd.so; no Python or torch in the measured process)"},
        "ppo_iter_ms": m["ppo_iter_ms"], "gym_steps_per_s": m["value"] / n_p, "physics_ticks_per_s": m["value"] / n_p * 8,
        "collect_ms_per_iter": m["ms_per_step"] - m["ppo_iter_ms"],
        "roofline": roof,
        "mfma": {"kernels": "rlgpu_ppo_minibatch: k_ppo_fwd_bwd (gather + forward + loss + dX chain of both networks) + k_dw_grouped (every dW / db), csrc/ppo_fused.h", "achieved_tflops": tflops, "peak_tflops": peak_tflops,
                 "ms_total": m["gemm_ms_total"], "calls": m["gemm_calls"], "frac": tflops / peak_tflops},
        # multi-GPU audit trail (bench_main): RCCL ranks that took part (0 = no communicator), each rank's own ms per iteration, one gradient all-reduce
        "rccl_ranks": m.get("rccl_ranks", 0), "rank_ms_per_step": m.get("rank_ms_per_step", []),
        "allreduce_ms_per_optimizer_step": m.get("allreduce_ms_per_optimizer_step", 0.0), "allreduce_calls": m.get("allreduce_calls", 0),
        # what carried the gradient exchange ("rccl"; "shm" = test transport; "none" = one rank) and every RLGPU_* switch the measured process saw
        "transport": transport, "env_overrides": m.get("env_overrides", []),
    }
    if "trained_regime" in m:
        out["trained_regime"] = dict(m["trained_regime"], note="same learner config (1 epoch) continued; at this point play is still close to random. "
                                     "A policy that has learned to chase the ball costs more per tick: the `trained_regime_learned` leg")
    if "trained_regime_learned" in m:
        out["trained_regime_learned"] = dict(m["trained_regime_learned"], note="the same learner after learning on with 2 PPO epochs per iteration until the policy plays "
                                             "(mean step reward and entropy next to the fresh policy's); measured with the headline's settings (1 epoch)")
    if args.mesh == "tessellated":
        out["config"]["workload"] = out["config"]["workload"].replace("procedural arena mesh", f"tessellated arena mesh ({mesh_info['triangles']} triangles in {mesh_info['files']} .cmf files)")
        out["mesh"] = mesh_info
    if not args.no_cpu_baseline and world == 1:
        cb = cpu_baseline(mesh_dir=mesh_dir if args.mesh == "tessellated" else None) if args.mesh != "tessellated" else tessellated_cpu_baseline(mesh_dir)
        if cb is not None:
            out["cpu_baseline"] = cb
    if world == 1 and not args.lockstep and m.get("collection") == "free-running":
        # the same measurement (shorter) with every env taking exactly T steps per iteration: what the launch's slowest wavefront costs
        cmd3 = [exe, "--envs", str(args.envs), "--team-size", str(args.team_size), "--horizon", str(args.horizon), "--steps", str(max(20, args.steps // 4)), "--warmup", str(max(5, args.warmup // 2)),
                "--epochs", str(args.epochs), "--lockstep"]
        if args.padded_zero_sum: cmd3.append("--padded-zero-sum")
        if args.fp32: cmd3.append("--fp32")
        if args.fp16: cmd3.append("--fp16")
        p3 = subprocess.run(cmd3, stdout=subprocess.PIPE, env=env, cwd=ROOT)
        if p3.returncode == 0:
            m3 = json.loads(p3.stdout.decode().strip().splitlines()[-1])
            out["lockstep_collection"] = {"value": m3["value"], "unit": "agent-steps/s", "ms_per_step": m3["ms_per_step"], "ppo_iter_ms": m3["ppo_iter_ms"], "steps": m3["steps"],
                                          "env_kernel_avg_ms": m3["env_kernel_ms_total"] / max(1, m3["env_launches"]), "gym_steps_per_launch": m3["gym_steps_per_launch"]}
    if args.mesh == "both" and world == 1:
        # the same measurement (shorter) on the tessellated mesh, with its own CPU baseline: both sides load the same 16 files
        mesh_dir, mesh_info = make_tessellated_mesh_dir()
        cmd2 = [exe, "--envs", str(args.envs), "--team-size", str(args.team_size), "--horizon", str(args.horizon), "--steps", str(max(20, args.steps // 4)), "--warmup", str(max(5, args.warmup // 2)),
                "--epochs", str(args.epochs), "--mesh-dir", mesh_dir]
        if args.padded_zero_sum: cmd2.append("--padded-zero-sum")
        if args.fp32: cmd2.append("--fp32")
        if args.fp16: cmd2.append("--fp16")
        p2 = subprocess.run(cmd2, stdout=subprocess.PIPE, env=env, cwd=ROOT)
        if p2.returncode == 0:
            m2 = json.loads(p2.stdout.decode().strip().splitlines()[-1])
            leg = dict(mesh_info, value=m2["value"], unit="agent-steps/s", ms_per_step=m2["ms_per_step"], ppo_iter_ms=m2["ppo_iter_ms"], steps=m2["steps"],
                       env_kernel_avg_ms=m2["env_kernel_ms_total"] / max(1, m2["env_launches"]), fused_collect=m2["fused_collect"])
            if not args.no_cpu_baseline:
                cb2 = tessellated_cpu_baseline(mesh_dir)
                if cb2 is not None:
                    leg["cpu_baseline"] = cb2
            out["mesh_tessellated"] = leg
            if ckpt_dir is not None:
                # ... and the same mesh under the policy the `trained_regime_learned` leg learned (loaded from its checkpoint; 40 iterations first, so that the
                # episodes in flight are that policy's): learned play AND a game-like mesh, the combination a real user runs
                cmd4 = [exe, "--envs", str(args.envs), "--team-size", str(args.team_size), "--horizon", str(args.horizon), "--steps", str(max(20, args.steps // 4)), "--warmup", "40",
                        "--epochs", str(args.epochs), "--mesh-dir", mesh_dir, "--load-checkpoint", ckpt_dir]
                p4 = subprocess.run(cmd4, stdout=subprocess.PIPE, env=env, cwd=ROOT)
                if p4.returncode == 0:
                    m4 = json.loads(p4.stdout.decode().strip().splitlines()[-1])
                    out["mesh_tessellated"]["learned_policy"] = {"value": m4["value"], "unit": "agent-steps/s", "ms_per_step": m4["ms_per_step"], "ppo_iter_ms": m4["ppo_iter_ms"], "steps": m4["steps"],
                                                                 "env_kernel_avg_ms": m4["env_kernel_ms_total"] / max(1, m4["env_launches"]),
                                                                 "mean_step_reward": m4.get("mean_step_reward"), "policy_entropy": m4.get("policy_entropy"),
                                                                 "note": "the policy of `trained_regime_learned` (its checkpoint), 40 iterations of play first, measured with the headline's settings"}
                else:
                    out["mesh_tessellated"]["learned_policy"] = {"error": f"bench_main exit code {p4.returncode}"}
        else:
            out["mesh_tessellated"] = {"error": f"bench_main exit code {p2.returncode}"}
    if ckpt_dir is not None:
        import shutil
        shutil.rmtree(ckpt_dir, ignore_errors=True)
    if world == 1 and not args.no_user_reward_leg and args.team_size == 1 and not args.padded_zero_sum:
        # what every RLGym user writes: a RewardFunction class of their own.  It has no device form, so it runs on the host -- after the fused collection launch
        # (LearnerConfig::deferHostRewards), on the host's threads
        cmd5 = [exe, "--envs", str(args.envs), "--horizon", str(args.horizon), "--steps", str(max(20, args.steps // 4)), "--warmup", str(max(5, args.warmup // 2)), "--epochs", str(args.epochs), "--user-reward"]
        p5 = subprocess.run(cmd5, stdout=subprocess.PIPE, env=env, cwd=ROOT)
        if p5.returncode == 0:
            m5 = json.loads(p5.stdout.decode().strip().splitlines()[-1])
            out["user_reward"] = {"value": m5["value"], "unit": "agent-steps/s", "ms_per_step": m5["ms_per_step"], "ppo_iter_ms": m5["ppo_iter_ms"], "steps": m5["steps"], "what": m5.get("user_reward"),
                                  "host_threads": m5.get("host_threads"), "host_cores": m5.get("host_cores"), "fused_collect": m5.get("fused_collect"), "collection": m5.get("collection")}
        else:
            out["user_reward"] = {"error": f"bench_main exit code {p5.returncode}"}
    if world == 1 and not args.no_config_legs and args.team_size == 1 and not args.padded_zero_sum:
        # BASELINE configs[3] and configs[4] AS WORDED, short legs outside the timed region: every number here is one bench_main run
        legs = {"c3": (["--team-size", "2", "--envs", "8192", "--padded-zero-sum", "--steps", "16", "--warmup", "4"],
                       "configs[3]: 2v2, 8192 envs/GPU, DefaultOBSPadded(2) shuffled + ZeroSumReward around the example stack, bf16 operands"),
                "c4": (["--team-size", "3", "--envs", "16384", "--padded-zero-sum", "--overlap", "--fp16", "--steps", "12", "--warmup", "3"],
                       "configs[4]: 3v3, 16384 envs/GPU, collect-during-learn overlap + fp16 operands with the dynamic loss scale, padded obs + zero-sum")}
        out["configs"] = {}
        for key, (extra, what) in legs.items():
            pc = subprocess.run([exe, "--horizon", str(args.horizon), "--epochs", str(args.epochs)] + extra, stdout=subprocess.PIPE, env=env, cwd=ROOT)
            if pc.returncode != 0:
                out["configs"][key] = {"workload": what, "error": f"bench_main exit code {pc.returncode}"}
                continue
            mc = json.loads(pc.stdout.decode().strip().splitlines()[-1])
            out["configs"][key] = {"workload": what, "value": mc["value"], "unit": "agent-steps/s", "ms_per_step": mc["ms_per_step"], "ppo_iter_ms": mc["ppo_iter_ms"], "steps": mc["steps"],
                                   "envs_per_gpu": mc["envs_per_gpu"], "team_size": mc["team_size"], "batch": mc["batch"], "operands": mc.get("operands"),
                                   "collection": mc.get("collection"), "collection_during_learn": mc.get("collection_during_learn"), "fused_collect": mc.get("fused_collect"),
                                   "env_kernel_avg_ms": mc["env_kernel_ms_total"] / max(1, mc["env_launches"])}
    print(json.dumps(out))


def make_tessellated_mesh_dir():
    """The procedural arena at ~9 k triangles (fillets of 8 strips, no edge longer than 700 uu) written as 16 .cmf files into a fresh
    directory <tmp>/soccar/ -- host code only (librlgpu.so's mesh generator), no GPU."""
    import tempfile
    from rlgymppo_cpp_amd.env import procedural_mesh_ex, write_cmf_files
    v, t = procedural_mesh_ex(8, 700.0)
    d = tempfile.mkdtemp(prefix="rlgpu_mesh_")
    paths = write_cmf_files(v, t, os.path.join(d, "soccar"), 16)
    return d, {"triangles": int(len(t)), "files": len(paths), "generator": "rlgpu_procedural_mesh_ex(fillet_segments=8, max_edge_uu=700), split by sector into .cmf files"}


def tessellated_cpu_baseline(mesh_dir):
    """The reference on the same files, in a process of its own (RocketSim initialises once per process)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", mesh_dir], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300, cwd=ROOT)
        line = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{") or ln == "null"]
        return json.loads(line[-1]) if line else None
    except Exception as e:
        return {"value": None, "unit": "agent-steps/s", "kind": "error", "sample": str(e)[:200]}


if __name__ == "__main__":
    main()
