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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--team-size", type=int, default=1, help="1 = BASELINE configs[1] (the headline); 2 / 3 with --padded-zero-sum = the shapes of configs[3] / [4]")
    ap.add_argument("--padded-zero-sum", action="store_true", help="DefaultOBSPadded(maxPlayers = team size) + ZeroSumReward around the example stack")
    ap.add_argument("--fp32", action="store_true", help="fp32 MFMA instead of bf16 operands")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true", help="LearnerConfig.collectionDuringLearn: the PPO epochs run on their own stream under the next collection (not the headline mode)")
    args = ap.parse_args()

    import torch
    from rlgymppo_cpp_amd import parallel
    rank, local_rank, world = parallel.init_process_group("nccl")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path is HIP kernels with no CPU fallback")

    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    n_agents = args.envs * 2 * args.team_size
    B = n_agents * args.horizon
    gym_cfg = None
    if args.padded_zero_sum:
        from rlgymppo_cpp_amd import _lib
        gym_cfg = _lib.default_gym_config(); gym_cfg.obs_max_players = args.team_size; gym_cfg.zero_sum = 1; gym_cfg.team_spirit = 0.3; gym_cfg.opp_scale = 1.0
    cfg = LearnerConfig(numEnvs=args.envs, teamSize=args.team_size, timestepsPerIteration=B, expBufferSize=B, device=local_rank, randomSeed=123, collectionDuringLearn=args.overlap,
                        ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 4, epochs=args.epochs, policyLR=2e-4, criticLR=2e-4, entCoef=0.01,
                                             autocastLearn=not args.fp32))
    L = Learner(cfg, gym_cfg=gym_cfg, rank=rank, world_size=world)

    def barrier():
        torch.cuda.synchronize()
        parallel.barrier(world)
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        L.iteration()
    barrier()
    L.env.timing_total(reset=True); L.ppo.timing_total(reset=True)
    cs = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        with torch.cuda.stream(L.s_collect):
            L.collect()
            c0 = torch.cuda.Event(enable_timing=True); c1 = torch.cuda.Event(enable_timing=True)
            c0.record()   # the library launches on the null stream, which is torch's current stream here
            L.add_new_experience()
            L.learn()
            c1.record()
        cs.append((c0, c1))
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = parallel.max_over_ranks(elapsed, world, torch.device("cuda", local_rank))
    consume_ms = sum(a.elapsed_time(b) for a, b in cs) / max(1, len(cs))
    env_ms, env_launches = L.env.timing_total(reset=False)
    gemm_ms, gemm_flops, gemm_calls = L.ppo.timing_total(reset=False)

    if rank == 0:
        agent_steps = B * world * args.steps
        value = agent_steps / elapsed
        n_p = 2 * args.team_size
        if L._fused_collect:
            # one launch = the whole collection phase: the resident state is read and written ONCE, every step writes its experience rows
            # (obs, reward, done, action, log-prob) and reads its observation rows back for the in-kernel inference
            per_launch_bytes = (2 * (336 * n_p + 264) + args.horizon * (n_p * (4 * L.obs_size + 8) + 4 + n_p * (4 * L.obs_size + 8))) * args.envs
        else:
            per_launch_bytes = algorithmic_bytes_per_gym_step(n_p, L.obs_size) * args.envs
        avg_launch_s = (env_ms / max(1, env_launches)) * 1e-3
        achieved = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        peak = 8000.0
        out = {
            "metric": "env steps/sec/node + PPO iter ms, 1v1 4096 envs/GPU", "value": value, "unit": "agent-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32 stepper + " + ("fp32" if args.fp32 else "bf16") + " MFMA MLP", "data": "synthetic",
            "config": {"workload": "%s, %d envs/GPU, tickSkip 8, %s(%d)+%sexample reward stack, RandomState resets, "
                                   "T=%d steps/iter, B=%d agent-steps/GPU, minibatch %d, epochs %d, MLP 256x3 policy(90)+critic, %s arena mesh"
                                   % ("BASELINE config[1]: 1v1" if args.team_size == 1 and not args.padded_zero_sum else "%dv%d (shape of BASELINE configs[%d])" % (args.team_size, args.team_size, args.team_size + 1),
                                      args.envs, "DefaultOBSPadded" if args.padded_zero_sum else "DefaultObs", L.obs_size, "zero-sum " if args.padded_zero_sum else "",
                                      args.horizon, B, B // 4, args.epochs, L.env.mesh_kind),
                       "envs_per_gpu": args.envs, "horizon": args.horizon, "batch": B, "minibatch": B // 4, "epochs": args.epochs},
            "collection_during_learn": bool(args.overlap),
            "ppo_iter_ms": consume_ms if not args.overlap else None, "gym_steps_per_s": value / (2 * args.team_size), "physics_ticks_per_s": value / (2 * args.team_size) * 8,
            "collect_ms_per_iter": (elapsed / args.steps * 1e3 - consume_ms) if not args.overlap else None,
            "roofline": {"kernel": ("k_env_collect<%d> (%d x (policy inference + 8 ticks + snapshot/obs/reward/done/auto-reset) in one launch)" % (2 * args.team_size, args.horizon)) if L._fused_collect else ("k_env_step<%d> (8 fused ticks + snapshot/obs/reward/done/auto-reset)" % (2 * args.team_size)), "bound": "hbm", "achieved": achieved, "peak": peak,
                         "unit": "GB/s", "frac": achieved / peak, "traffic": None, "avg_launch_ms": env_ms / max(1, env_launches), "launches": env_launches,
                         "algorithmic_bytes_per_launch": per_launch_bytes},
            "mfma": {"kernels": "k_gemm fwd+bwd of policy and critic inside rlgpu_ppo_minibatch (incl. loss kernels)", "achieved_tflops": (gemm_flops / (gemm_ms * 1e-3) / 1e12) if gemm_ms > 0 else 0.0,
                     "peak_tflops": 157.3 if args.fp32 else 2500.0, "ms_total": gemm_ms, "calls": gemm_calls},
        }
        out["mfma"]["frac"] = out["mfma"]["achieved_tflops"] / out["mfma"]["peak_tflops"]
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
