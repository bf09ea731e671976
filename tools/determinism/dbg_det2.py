import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
team, n_envs, use_bf16, W, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == '1', int(sys.argv[4]), sys.argv[5]
dev = torch.device("cuda", 0)
CAP = 12
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
ea = BatchedEnv(n_envs, team, cfg)
core = PPOCore(ea.obs_size, ea.n_actions, (64, 64), (64, 64), use_bf16=use_bf16, max_rows=4096)
N, D = ea.n_agents, ea.obs_size
A = (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
     torch.full((CAP, N), -777.0, device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
torch.cuda.synchronize()
import os, ctypes
if os.environ.get('SCRUB'):
    sl = ctypes.CDLL(os.path.join('' + os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'probes', 'libscrub_probe.so') + '')); print('scrub', sl.scrub_probe(ctypes.c_uint(int(os.environ['SCRUB'], 0))))
ea.reset(True, A[0][0])
res = {}
for k in range(W + 1):
    if os.environ.get('SCRUB'): sl.scrub_probe(ctypes.c_uint(int(os.environ['SCRUB'], 0)))
    assert ea.collect(core, CAP, *A); ea.sync()
    for nm, x in zip(("obs", "act", "logp", "rew", "done"), A): res[f"{nm}{k}"] = x.cpu().numpy().copy()
    A[0][0].copy_(A[0][CAP]); torch.cuda.synchronize()
np.savez(out, **res)
