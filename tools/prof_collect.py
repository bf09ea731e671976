"""Profiler build only (RLGPU_LIB=.../librlgpu_prof.so): per-workgroup cycles of the fused collection kernel (rlgpu_collect)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
n, T = 4096, 32
env = BatchedEnv(n, 1); dev = torch.device("cuda", 0)
N, D = env.n_agents, env.obs_size
ppo = PPOCore(D, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=65536, seed=1)
obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
env.reset(True, obs[0])
fn = env.lib.rlgpu_env_debug_step_prof; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; fn.restype = C.c_int
nb = (n + 3) // 4
for it in range(12):
    assert env.collect(ppo, T, obs, acts, logp, rew, done)
    obs[0].copy_(obs[T])
    buf = np.zeros(16 * nb, dtype=np.uint64)
    assert fn(env.h, buf.ctypes.data, nb) == 0
    tot = buf.reshape(-1, 16)[:, 0].astype(np.float64); inf = buf.reshape(-1, 16)[:, 1].astype(np.float64); mlp = buf.reshape(-1, 16)[:, 2].astype(np.float64)
    if it >= 9:
        q = np.percentile(tot, [1, 50, 90, 99])
        print("launch %d: workgroup cycles min %.2fM p1 %.2fM median %.2fM mean %.2fM p90 %.2fM p99 %.2fM max %.2fM (%.1f ms); inference share of the mean %.1f%% (per step: MLP %.0fK cycles, head %.0fK); mean/max %.2f"
              % (it, tot.min() / 1e6, q[0] / 1e6, q[1] / 1e6, tot.mean() / 1e6, q[2] / 1e6, q[3] / 1e6, tot.max() / 1e6, tot.max() / 2.38e6, 100 * inf.mean() / tot.mean(), mlp.mean() / T / 1e3, (inf.mean() - mlp.mean()) / T / 1e3, tot.mean() / tot.max()))
