"""Do kernels of two HIP streams run concurrently on this box?  32 env steps (null stream, LDS-saturating, long tail) against a chain
of bf16 GEMMs of about the same total time on a side stream: both-at-once close to the max of the two = overlap, close to the sum = none."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
env = BatchedEnv(4096, 1)
dev = torch.device("cuda", 0)
obs = env.reset(True)
nobs = torch.empty_like(obs); rew = torch.empty(env.n_agents, device=dev); done = torch.empty(env.n_agents, dtype=torch.int32, device=dev)
acts = torch.randint(0, 90, (64, env.n_agents), dtype=torch.int32).to(dev)
a = torch.randn(65536, 256, device=dev, dtype=torch.bfloat16); w = torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
side = torch.cuda.Stream()
def envs(n=32):
    for t in range(n): env.step(acts[t % 64], nobs, rew, done)
def gemms(n):
    x = a
    for _ in range(n): x = torch.relu(x @ w)
def timed(f):
    torch.cuda.synchronize(); t = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3
for _ in range(3): envs(8); gemms(50)
t_env = timed(envs)
n_g = 200
t_g = timed(lambda: gemms(n_g))
n_g = max(10, int(n_g * t_env / t_g))
t_g = timed(lambda: gemms(n_g))
def both():
    envs()
    with torch.cuda.stream(side): gemms(n_g)
t_both = timed(both)
print("env alone %.2f ms, %d gemms alone %.2f ms, both %.2f ms (sum %.2f, max %.2f)" % (t_env, n_g, t_g, t_both, t_env + t_g, max(t_env, t_g)))
