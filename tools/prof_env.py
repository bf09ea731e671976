"""Small driver for rocprofv3: N env steps of the BASELINE config with random actions (no learner)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
env = BatchedEnv(n, 1)
dev = torch.device("cuda", 0)
obs = env.reset(True)
nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
g = torch.Generator().manual_seed(0)
for t in range(steps):
    a = torch.randint(0, 90, (env.n_agents,), generator=g, dtype=torch.int32).to(dev)
    env.step(a, nobs, rew, done)
env.sync()
ms, k = env.timing_total()
print("avg step kernel ms", ms / k, "launches", k)
