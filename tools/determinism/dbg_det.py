import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
team, n_envs = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
ea, eb = BatchedEnv(n_envs, team, cfg), BatchedEnv(n_envs, team, cfg)
N, D = ea.n_agents, ea.obs_size
oa = torch.zeros((N, D), device=dev); ob = torch.zeros((N, D), device=dev); torch.cuda.synchronize()
ea.reset(True, oa); eb.reset(True, ob); ea.sync(); eb.sync()
sa, sb = ea.download_states(), eb.download_states()
print('after reset: states equal', all(bytes(x) == bytes(y) for x, y in zip(sa, sb)), 'obs equal', bool((oa == ob).all()))
g = torch.Generator().manual_seed(0)
nobs_a = torch.empty_like(oa); nobs_b = torch.empty_like(ob); r = torch.empty(N, device=dev); d = torch.empty(N, dtype=torch.int32, device=dev)
r2 = torch.empty(N, device=dev); d2 = torch.empty(N, dtype=torch.int32, device=dev)
for t in range(40):
    a = torch.randint(0, 90, (N,), generator=g, dtype=torch.int32).to(dev); torch.cuda.synchronize()
    ea.step(a, nobs_a, r, d); eb.step(a, nobs_b, r2, d2); ea.sync(); eb.sync()
    if not bool((nobs_a == nobs_b).all()) or not bool((r == r2).all()):
        bad = (nobs_a != nobs_b).nonzero()[:4].tolist()
        print('step', t, 'obs differ at', bad, 'rewards differ', int((r != r2).sum()), 'dones', int(d.sum()), int(d2.sum()))
        sa, sb = ea.download_states(), eb.download_states()
        for e, (x, y) in enumerate(zip(sa, sb)):
            if bytes(x) != bytes(y):
                bx, by = np.frombuffer(bytes(x), np.uint8), np.frombuffer(bytes(y), np.uint8)
                print(' env', e, 'first differing byte', int(np.argmax(bx != by)), 'of', len(bx), 'n diff', int((bx != by).sum())); break
        break
else:
    print('40 steps equal')
