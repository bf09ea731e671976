import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0); CAP = 12
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
core = PPOCore(89, 90, (64, 64), (64, 64), use_bf16=False, max_rows=4096)
stream, ctr = core.get_sampler()
runs = []
for r in range(3):
    env = BatchedEnv(n_envs, 1, cfg); N, D = env.n_agents, env.obs_size
    A = (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
         torch.full((CAP, N), -777.0, device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
    torch.cuda.synchronize(); core.set_sampler(stream, ctr)
    env.reset(True, A[0][0]); assert env.collect(core, CAP, *A); env.sync()
    runs.append([x.cpu().numpy().copy() for x in A]); env.close()
for r in (1, 2):
    for nm, a, b in zip(("obs", "act", "logp", "rew", "done"), runs[0], runs[r]):
        bad = a != b
        if bad.any():
            idx = np.argwhere(bad)
            print(f"run {r} vs 0: {nm} differs in {int(bad.sum())}, first {idx[0].tolist()}, envs {sorted(set((idx[:, 1] // 2).tolist()))[:10]}")
bad = runs[0][3] != runs[1][3]
if bad.any():
    t, row = np.argwhere(bad)[0]; e = row // 2
    print("env", e, "first reward difference at step", t)
    for r in range(3):
        print(" run", r, "rewards rows", runs[r][3][max(0, t - 2):t + 2, 2 * e:2 * e + 2].tolist(), "done", runs[r][4][max(0, t - 2):t + 2, 2 * e].tolist(), "actions", runs[r][1][max(0, t - 2):t + 1, 2 * e:2 * e + 2].tolist())
        print("   obs[t] self row0 first 12:", np.round(runs[r][0][t, 2 * e, :12], 5).tolist())
        print("   obs[t+1]             :", np.round(runs[r][0][t + 1, 2 * e, :12], 5).tolist())
