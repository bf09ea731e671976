"""Physics-only tick rate of the stepper (rlgpu_env_physics_ticks) after a random-action warm-up: env-ticks per second.
   usage: tick_rate.py [envs] [warm-up steps] [team size]      (A/B runs of stepper variants through RLGPU_LIB, tools/build_variant.sh)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = BatchedEnv(n, ts)
dev = torch.device("cuda", 0)
obs = env.reset(True)
g = torch.Generator(device=dev); g.manual_seed(3)
nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
for _ in range(warm):
    acts = torch.randint(0, env.n_actions, (env.n_agents,), generator=g, device=dev, dtype=torch.int32)
    env.step(acts, nobs, rew, done)
torch.cuda.synchronize()
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); env.physics_ticks(64); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
print(f"{os.environ.get('RLGPU_LIB', 'product')} dyn_lds={os.environ.get('RLGPU_EXPERIMENT_DYN_LDS', 0)}: {n} envs x 64 ticks in {best*1e3:.2f} ms = {n*64/best/1e6:.1f} M env-ticks/s")
