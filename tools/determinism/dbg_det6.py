import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
n_envs = int(sys.argv[1]); bf16 = sys.argv[2] == '1'; T = 12
dev = torch.device("cuda", 0)
out = []
core = PPOCore(89, 90, (64, 64), (64, 64), use_bf16=bf16, max_rows=4096)
stream, ctr = core.get_sampler()
for fused in (False, True):
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
    env = BatchedEnv(n_envs, 1, cfg); N, D = env.n_agents, env.obs_size
    obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev)
    logp = torch.zeros((T, N), device=dev); rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); core.set_sampler(stream, ctr)
    env.reset(True, obs[0])
    if fused: assert env.collect(core, T, obs, acts, logp, rew, done)
    else:
        for t in range(T):
            core.act(obs[t], acts[t], logp[t]); core.sync(); env.step(acts[t], obs[t + 1], rew[t], done[t]); env.sync()
    env.sync()
    out.append([x.cpu().numpy() for x in (obs, acts, rew, done)]); env.close()
for nm, a, b in zip(("obs", "act", "rew", "done"), out[0], out[1]):
    bad = a != b
    if bad.any():
        idx = np.argwhere(bad)
        print(nm, "collect != alternating in", int(bad.sum()), "first", idx[0].tolist(), "envs", sorted(set((idx[:, 1] // 2).tolist()))[:12])
print("done steps of env 172:", out[0][3][:, 344].tolist())
if len(sys.argv) > 3: np.savez(sys.argv[3], **{f"{k}{i}": out[i][j] for i in range(2) for j, k in enumerate(("obs", "act", "rew", "done"))})
