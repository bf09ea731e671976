"""In-kernel clocks of the physics tick: per-workgroup shader cycles and wall time (s_memtime / s_memrealtime)."""
import os, sys, ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import default_arena

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 24   # random gym steps before the rollout measurements (300+ = steady state of episodes)
env = BatchedEnv(n, 1)
fn = env.lib.rlgpu_env_debug_tick_cycles
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
fn.restype = C.c_int


def run(label):
    buf = np.zeros(10 * 65536, dtype=np.uint64)
    nb = C.c_int()
    for _ in range(3):
        rc = fn(env.h, ticks, buf.ctypes.data, 65536, C.byref(nb))
        assert rc == 0, rc
    b = buf[: 10 * nb.value].reshape(-1, 10).astype(np.float64)
    cyc, rt = b[:, 0], b[:, 1]
    us = rt / 100.0  # s_memrealtime: 100 MHz
    print(f"{label}: blocks {nb.value}  cycles/tick mean {cyc.mean()/ticks:.0f} min {cyc.min()/ticks:.0f} max {cyc.max()/ticks:.0f}  "
          f"us/tick mean {us.mean()/ticks:.1f} max {us.max()/ticks:.1f}  clock MHz {np.median(cyc/us):.0f}")
    ph = b[:, 2:].mean(axis=0) / ticks
    if ph.sum() > 0:
        names = ["car pre-tick+pads/gravity", "narrowphase walk", "narrowphase items", "contacts+solver setup", "solver iters", "integrate", "post/pads/ball", "wheel ray casts"]
        print("   phases (cycles/tick, mean over blocks): " + ", ".join(f"{n} {v:.0f}" for n, v in zip(names, ph) if v > 0))
        worst = int(np.argmax(cyc))
        print("   slowest block: " + ", ".join(f"{n} {v/ticks:.0f}" for n, v in zip(names, b[worst, 2:]) if v > 0))


for label, thr in (("rest", 0.0), ("throttle", 1.0)):
    s = default_arena(2)
    for k in range(2):
        s.cars[k].controls[0] = thr
    env.upload_states([s] * n)
    run(label)
obs = env.reset(True)
dev = torch.device("cuda", 0)
nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
g = torch.Generator().manual_seed(0)
for t in range(warm):
    a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
    env.step(a, nobs, rew, done)
env.sync()
run("random-rollout")

# whole gym step (profiler build only): the same buckets plus the non-tick parts of k_env_step
if hasattr(env.lib, "rlgpu_env_debug_step_prof"):
    fn2 = env.lib.rlgpu_env_debug_step_prof
    fn2.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; fn2.restype = C.c_int
    nb = (n + 3) // 4
    acc = np.zeros((min(nb, 4096), 16))
    reps = 8
    per_launch = []
    for t in range(reps):
        a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
        env.step(a, nobs, rew, done)
        buf = np.zeros(16 * min(nb, 4096), dtype=np.uint64)
        assert fn2(env.h, buf.ctypes.data, nb) == 0
        acc += buf.reshape(-1, 16).astype(np.float64)
        per_launch.append(buf.reshape(-1, 16)[:, :12].astype(np.float64).sum(axis=1))
    acc /= reps
    names = ["car pre-tick+pads/gravity", "candidates", "narrowphase items", "contacts+solver setup", "solver iters", "integrate", "post/pads/ball",
             "wheel ray casts", "load+parse actions", "tracker/snapshot/reward/done", "reset+obs", "store"]
    tot = acc[:, :12].sum(axis=1)
    pl = np.stack(per_launch)   # [launch][workgroup]
    q = np.percentile(pl, [50, 90, 99, 99.9], axis=1).mean(axis=1)
    print("per launch, over its workgroups: median %.0f  p90 %.0f  p99 %.0f  p99.9 %.0f  max %.0f (mean of %d launches); workgroups above 80%% of their launch's max: %.1f"
          % (q[0], q[1], q[2], q[3], pl.max(axis=1).mean(), reps, (pl > 0.8 * pl.max(axis=1, keepdims=True)).sum(axis=1).mean()))
    print(f"k_env_step cycles per launch: mean {tot.mean():.0f} max {tot.max():.0f} ({tot.max()/2.38e3:.0f} us)")
    print("   mean: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, acc[:, :12].mean(axis=0))))
    w = int(np.argmax(tot))
    print("   slowest wg: " + ", ".join(f"{nm} {v:.0f}" for nm, v in zip(names, acc[w, :12])))

if hasattr(env.lib, "rlgpu_env_debug_ints"):
    buf = (C.c_int * 64)(); env.lib.rlgpu_env_debug_ints.argtypes = [C.c_void_p, C.c_void_p]; env.lib.rlgpu_env_debug_ints(env.h, buf)
    print("queue overflow events so far: frontier", buf[0], "ball region", buf[1], "car region", buf[2], "items", buf[3], "pool", buf[4])

    print("hitbox-triangle GJK runs", buf[8], " of them answered by the deep-penetration fallback", buf[9])
    if buf[11]:
        print(f"in-wavefront spans: GJK {64.0 * buf[10] / buf[11]:.0f} cycles per run ({buf[11]} runs), internal-edge adjustment {64.0 * buf[12] / max(1, buf[13]):.0f} cycles per contact ({buf[13]})"
              " -- wall cycles of the wavefront while the lane was inside, i.e. including lanes of other items diverged into the same call")
    if buf[34]:
        print(f"GJK: {buf[32]} hitbox-triangle tests, {buf[33]} runs, iterations per run mean {buf[35] / buf[34]:.2f} max {buf[36]}; simplex updates with 1..4 vertices {list(buf[37:41])}; "
              f"runs by iteration count 0..22+: {list(buf[41:64])}")
    for t, nm in enumerate(("ball-triangle", "hitbox-triangle", "car-car")):
        c = buf[16 + 4 * t]
        if c: print(f"items {nm}: {c} run, {buf[17 + 4 * t]} with a contact, mean {64.0 * buf[18 + 4 * t] / c:.0f} cycles, slowest {buf[19 + 4 * t]}")
