"""per-workgroup cycles of a LOCKSTEP collection launch with more workgroups than the device holds: how far is the launch from sum / slots?"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
team, n, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = _lib.default_gym_config()
env = BatchedEnv(n, team, cfg); dev = torch.device("cuda", 0)
N, D = env.n_agents, env.obs_size
ppo = PPOCore(D, 90, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=65536, seed=1)
obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev); logp = torch.zeros((T, N), device=dev)
rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
env.reset(True, obs[0])
fn = env.lib.rlgpu_env_debug_step_prof; fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]; fn.restype = C.c_int
epw = {1: 4, 2: 3, 3: 2}[team]
nb = min(4096, (n + epw - 1) // epw)
for it in range(8):
    torch.cuda.synchronize(); t0 = time.time()
    assert env.collect(ppo, T, obs, acts, logp, rew, done); env.sync()
    torch.cuda.synchronize(); ms = (time.time() - t0) * 1e3
    obs[0].copy_(obs[T])
    buf = np.zeros(16 * nb, dtype=np.uint64); assert fn(env.h, buf.ctypes.data, nb) == 0
    tot = buf.reshape(-1, 16)[:, 0].astype(np.float64)
    if it >= 4:
        clk = 2.38e9
        print("launch %d: %d workgroups, launch %.1f ms; workgroup time mean %.2f ms max %.2f ms; sum / 1024 slots = %.1f ms (%.0f %% of the launch)" % (it, nb, ms, tot.mean() / clk * 1e3, tot.max() / clk * 1e3, tot.sum() / 1024 / clk * 1e3, 100 * tot.sum() / 1024 / clk * 1e3 / ms))
