"""Tick-only workload for rocprofv3 --pmc passes: N launches of k_env_ticks (8 ticks each) on the rest state, or on the arenas a random
   rollout of `warm` gym steps has left.  usage: tick_pmc.py rest|random [envs] [launches] [warm]
   Read the per-kernel counter sums with tools/read_prof.py; per tick and wavefront = sum / (launches * 8 * envs / 4)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import default_arena
mode = sys.argv[1] if len(sys.argv) > 1 else "rest"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 10
warm = int(sys.argv[4]) if len(sys.argv) > 4 else 300
env = BatchedEnv(n, 1)
dev = torch.device("cuda", 0)
if mode == "rest":
    env.upload_states([default_arena(2)] * n)
else:
    obs = env.reset(True)
    nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
    g = torch.Generator().manual_seed(0)
    for t in range(warm):
        a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
        env.step(a, nobs, rew, done)
env.physics_ticks(8); env.sync()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(launches):
    env.physics_ticks(8)
e1.record(); torch.cuda.synchronize()
print(mode, "ms per 8 ticks:", e0.elapsed_time(e1) / launches, "-> us per tick", e0.elapsed_time(e1) / launches / 8 * 1e3)
