import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
dev = torch.device("cuda", 0)
res = []
for fused in (False, True):
    cfg = _lib.default_gym_config(); cfg.seed_lo = 31
    env = BatchedEnv(64, 1, cfg=cfg); N, D = env.n_agents, env.obs_size
    ppo = PPOCore(D, 90, (256, 256, 256), (64,), use_bf16=True, max_rows=128, seed=5)
    T = 1
    obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev)
    logp = torch.zeros((T, N), device=dev); rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
    env.reset(True, obs[0])
    if fused: env.collect(ppo, T, obs, acts, logp, rew, done)
    else:
        ppo.act(obs[0], acts[0], logp[0]); env.step(acts[0], obs[1], rew[0], done[0])
    env.sync()
    res.append((acts.cpu().numpy(), logp.cpu().numpy(), ppo.probs(obs[0]).cpu().numpy()))
a0, l0, p0 = res[0]; a1, l1, p1 = res[1]
print("actions equal", (a0 == a1).all(), "logp diff rows", np.nonzero(l0[0] != l1[0])[0][:20], "max", np.abs(l0 - l1).max())
want = np.log(p0[np.arange(p0.shape[0]), a0[0]])
print("seq logp vs log(probs) maxdiff", np.abs(l0[0] - want).max(), " fused:", np.abs(l1[0] - want).max())
