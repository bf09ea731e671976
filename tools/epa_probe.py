"""How often the penetration-depth solver runs and what a collection launch costs: 12 fused collection launches of the headline config from a
fresh policy; prints ms per launch, EPA queries (and those in the full-size arena) and queue overflows per million env-ticks."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
n, T = 4096, 32
mesh = "procedural"
if len(sys.argv) > 1 and sys.argv[1] == "tess":
    import bench
    mesh, info = bench.make_tessellated_mesh_dir(); mesh = os.path.join(mesh, "soccar"); print("mesh:", info)
env = BatchedEnv(n, 1, mesh=mesh); dev = torch.device("cuda", 0)
N, D = env.n_agents, env.obs_size
ppo = PPOCore(D, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=65536, seed=1)
obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
env.reset(True, obs[0])
for it in range(12):
    env.epa_counts(reset=True); env.overflow_counts(reset=True)
    torch.cuda.synchronize(); t0 = time.time()
    assert env.collect(ppo, T, obs, acts, logp, rew, done)
    torch.cuda.synchronize(); ms = (time.time() - t0) * 1e3
    obs[0].copy_(obs[T])
    e = env.epa_counts(); o = env.overflow_counts()
    ticks = n * T * 8 / 1e6
    print("launch %2d: %.2f ms  EPA queries %d (%.1f / M env-ticks), full-size arena %d, overflows %s" % (it, ms, e[0], e[0] / ticks, e[1], o))
