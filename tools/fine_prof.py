"""Per-phase cycles of arena_tick_wave, one bucket per phase (needs a build with -DRLG_TICK_PROFILE -DRLG_FINE_PROF:
   make -C rlgymppo_cpp_amd/csrc prof EXTRA=-DRLG_FINE_PROF, loaded through RLGPU_LIB).  Sums over all workgroups; printed per workgroup and tick.
   usage: fine_prof.py [envs] [random warm-up steps] [learner iterations]: with the third argument the arenas are profiled where a policy
   trained for that many iterations (tools/train_probe.py's configuration) has left them, instead of after a random rollout."""
import os, sys, ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import default_arena

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
trained = int(sys.argv[3]) if len(sys.argv) > 3 else 0
mesh = "procedural"
if len(sys.argv) > 4 and sys.argv[4] == "tess":      # the arena at ~10 k triangles in 16 .cmf files (bench.py --mesh tessellated)
    import bench
    mesh, info = bench.make_tessellated_mesh_dir(); mesh = os.path.join(mesh, "soccar"); print("mesh:", info)
ticks, reps = 8, 4
if trained:
    from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
    B = n * 2 * 32
    L = Learner(LearnerConfig(numEnvs=n, teamSize=1, timestepsPerIteration=B, expBufferSize=B, randomSeed=1,
                              ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 4, epochs=2, policyLR=2e-4, criticLR=2e-4, entCoef=0.01, autocastLearn=True)))
    env = L.env
else:
    env = BatchedEnv(n, 1, mesh=mesh)
fn = env.lib.rlgpu_env_debug_tick_cycles
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]; fn.restype = C.c_int
env.lib.rlgpu_env_debug_ints.argtypes = [C.c_void_p, C.c_void_p]
SUB = {32: "car2: copy-in + wheel flags", 33: "car2: update_wheels", 34: "car2: air/jump/flip/roll", 35: "car2: suspension impulses", 36: "car2: friction impulses + boost",
       37: "cand: query boxes + stale vote", 38: "cand: walk", 39: "cand: kept leaves copy", 40: "wheel1c: load + mesh key + ball/cars", 41: "wheel1c: suspension",
       42: "wheel1c: friction impulse", 43: "body: bp_history_cell", 44: "prepare: collide_merge", 45: "prepare: solver bodies", 46: "finish: velocities + position",
       47: "wheel1a: basis", 48: "finish: integrate_rotation"}
NAMES = ["car_tick_begin", "build_candidates", "wheel_ray_begin", "ray pairs", "wheel_ray_finish", "car_pre_tick_finish", "pads + world_begin",
         "candidate tests", "items", "body contacts", "solver_prepare", "solver_rows", "solver_iterate", "solver_finish", "car post + pad check",
         "pads_lock", "pad post", "tick_finish"]


def ints():
    buf = (C.c_int * 64)(); env.lib.rlgpu_env_debug_ints(env.h, buf); return np.array(buf[:64], dtype=np.int64)


def run(label):
    buf = np.zeros(10 * 65536, dtype=np.uint64); nb = C.c_int()
    fn(env.h, ticks, buf.ctypes.data, 65536, C.byref(nb))   # warm
    a = ints()
    tot = np.zeros(0)
    for _ in range(reps):
        assert fn(env.h, ticks, buf.ctypes.data, 65536, C.byref(nb)) == 0
        tot = np.concatenate([tot, buf[: 10 * nb.value].reshape(-1, 10)[:, 0].astype(np.float64)])
        clk = buf[: 10 * nb.value].reshape(-1, 10)[:, 0].astype(np.float64) / np.maximum(1.0, buf[: 10 * nb.value].reshape(-1, 10)[:, 1].astype(np.float64)) * 0.1   # shader cycles per 100 MHz tick -> GHz
        print(f"   in-kernel clock (s_memtime / s_memrealtime): median {np.median(clk):.3f} GHz")
    d = (ints() - a) * 1024.0 / (reps * ticks * nb.value)
    print(f"{label}: cycles/tick mean {tot.mean()/ticks:.0f} max {tot.max()/ticks:.0f}; buckets sum {d.sum():.0f}")
    blk = np.zeros(32 * nb.value, dtype=np.uint32)
    env.lib.rlgpu_env_debug_fine.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    env.lib.rlgpu_env_debug_fine(env.h, blk.ctypes.data, nb.value)
    blk = blk.reshape(-1, 32).astype(np.float64) * 16.0 / ticks
    order = np.argsort(-blk.sum(axis=1))
    slow = blk[order[: max(1, nb.value // 100)]].mean(axis=0)   # the slowest 1 % of the workgroups of the last launch
    print(f"   {'phase':24s} {'mean':>8s}  {'share':>7s}  {'slowest 1 %':>12s}")
    for nm, v, w in zip(NAMES, d, slow):
        print(f"   {nm:24s} {v:8.0f}  {100*v/d.sum():5.1f} %  {w:12.0f}")
    for i in range(32, 64):      # sub-phase stamps (RLG_SPROF(i) inside the phase functions): the cycles up to stamp i since the stamp before it
        if d[i] > 0: print(f"   sub {i:2d} {SUB.get(i, ''):32s} {d[i]:8.0f}  {100*d[i]/d.sum():5.1f} %")
    print(f"   {'sum':24s} {d.sum():8.0f}           {slow.sum():12.0f}")


if trained:
    for it in range(trained):
        L.iteration()
    env.sync()
    run("after %d learner iterations" % trained)
    sys.exit(0)
s = default_arena(2)
env.upload_states([s] * n)
run("rest")
obs = env.reset(True)
dev = torch.device("cuda", 0)
nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
g = torch.Generator().manual_seed(0)
for t in range(warm):
    a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
    env.step(a, nobs, rew, done)
env.sync()
run("random-rollout")
