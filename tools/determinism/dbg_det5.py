import sys, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd import _lib
n_envs, out = int(sys.argv[1]), sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda", 0)
cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 9
env = BatchedEnv(n_envs, 1, cfg); N, D = env.n_agents, env.obs_size
obs = torch.zeros((N, D), device=dev); torch.cuda.synchronize()
env.reset(True, obs); env.sync()
g = torch.Generator().manual_seed(0)
res = {"obs_reset": obs.cpu().numpy().copy()}
nobs = torch.empty_like(obs); r = torch.empty(N, device=dev); d = torch.empty(N, dtype=torch.int32, device=dev)
for t in range(steps):
    a = torch.randint(0, 90, (N,), generator=g, dtype=torch.int32).to(dev); torch.cuda.synchronize()
    env.step(a, nobs, r, d); env.sync()
    res[f"obs{t:02d}"] = nobs.cpu().numpy().copy(); res[f"rew{t:02d}"] = r.cpu().numpy().copy(); res[f"done{t:02d}"] = d.cpu().numpy().copy()
st = env.download_states()
res["final"] = np.frombuffer(b"".join(bytes(s) for s in st), np.uint8).copy()
np.savez(out, **res)
