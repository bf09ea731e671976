"""Which envs make a step launch slow?  Profiler build: after a steady-state warm-up, one rlgpu_env_step; the workgroups with the most
cycles, their phase buckets, and what the cars / ball of their envs are doing (state BEFORE that step)."""
import os, sys, ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 300
top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
env = BatchedEnv(n, 1)
fn2 = env.lib.rlgpu_env_debug_step_prof
fn2.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; fn2.restype = C.c_int
obs = env.reset(True)
dev = torch.device("cuda", 0)
nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
g = torch.Generator().manual_seed(0)
for t in range(warm):
    a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
    env.step(a, nobs, rew, done)
env.sync()
names = ["pre-tick", "candidates", "items", "setup", "solver", "integrate", "post", "rays", "load", "tracker", "reset", "store"]
for rep in range(3):
    before = env.download_states()
    a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
    env.step(a, nobs, rew, done); env.sync()
    nb = (n + 3) // 4
    buf = np.zeros(16 * nb, dtype=np.uint64)
    assert fn2(env.h, buf.ctypes.data, nb) == 0
    b = buf.reshape(-1, 16)[:, :12].astype(np.float64)
    tot = b.sum(axis=1)
    order = np.argsort(-tot)[:top]
    print(f"launch {rep}: mean {tot.mean():.0f} max {tot.max():.0f}")
    for w in order:
        print(f"  wg {w}: total {tot[w]:.0f}  " + " ".join(f"{nm} {v/1000:.0f}K" for nm, v in zip(names, b[w]) if v > 20000))
        for e in range(4 * w, 4 * w + 4):
            s = before[e]
            desc = []
            for k in range(2):
                c = s.cars[k]
                f = c.flags
                desc.append(f"car{k} z {c.pos[2]:.0f} up.z {c.rot[8]:+.2f} |v| {np.linalg.norm(c.vel[:]):.0f} |w| {np.linalg.norm(c.ang_vel[:]):.1f} "
                            f"{'G' if f & 1 else '-'}{'W' if f & (1 << 12) else '-'}{'D' if f & (1 << 13) else '-'} wheels {(f >> 1) & 15:04b} x {c.pos[0]:.0f} y {c.pos[1]:.0f}")
            print(f"     env {e}: ball z {s.ball.pos[2]:.0f} |v| {np.linalg.norm(s.ball.vel[:]):.0f} | " + " | ".join(desc))
