import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
team, n_envs, use_bf16 = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == '1'
dev = torch.device("cuda", 0)
T = 6; CAP = 2 * T
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
ea, eb = BatchedEnv(n_envs, team, cfg), BatchedEnv(n_envs, team, cfg)
core = PPOCore(ea.obs_size, ea.n_actions, (64, 64), (64, 64), use_bf16=use_bf16, max_rows=4096)
N, D, P = ea.n_agents, ea.obs_size, ea.n_agents // n_envs
def bufs():
    return (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
            torch.full((CAP, N), -777.0, device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
A, Bf = bufs(), bufs()
steps = torch.full((n_envs,), -1, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
stream, ctr = core.get_sampler()
ea.reset(True, A[0][0])
W = int(sys.argv[4]) if len(sys.argv) > 4 else 3
for _ in range(W):
    assert ea.collect(core, CAP, *A); ea.sync(); A[0][0].copy_(A[0][CAP]); torch.cuda.synchronize()
assert ea.collect(core, CAP, *A); ea.sync()
core.set_sampler(stream, ctr)
eb.reset(True, Bf[0][0])
for _ in range(W):
    assert eb.collect(core, CAP, *Bf); eb.sync(); Bf[0][0].copy_(Bf[0][CAP]); torch.cuda.synchronize()
Bf[3].fill_(-777.0); torch.cuda.synchronize()
mode = sys.argv[5] if len(sys.argv) > 5 else 'free'
if mode == 'free':
    assert eb.collect_free(core, CAP, T * N, *Bf, steps)
    eb.sync(); st = steps.cpu().numpy()
else:
    assert eb.collect(core, CAP, *Bf); eb.sync(); st = np.full(n_envs, CAP)
print('steps min/max', st.min(), st.max())
mask = (np.arange(CAP)[:, None] < st[None, :])
for name, a, b in zip(("actions", "logp", "reward", "done"), A[1:], Bf[1:]):
    a = a.cpu().numpy().reshape(CAP, n_envs, P); b = b.cpu().numpy().reshape(CAP, n_envs, P)
    bad = ((a != b) & mask[:, :, None])
    print(name, 'mismatches', bad.sum(), 'first at (step, env, player)', np.argwhere(bad)[:5].tolist())
    if bad.any() and name == 'logp':
        i = np.argwhere(bad)[0]; print('  values', a[tuple(i)], b[tuple(i)])
oa = A[0].cpu().numpy().reshape(CAP + 1, n_envs, P * D); ob = Bf[0].cpu().numpy().reshape(CAP + 1, n_envs, P * D)
bad = ((oa != ob) & (np.arange(CAP + 1)[:, None] <= st[None, :])[:, :, None])
print('obs mismatches', bad.sum(), np.argwhere(bad)[:3].tolist())
