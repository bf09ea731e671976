"""Why does the 2v2 / 21-env fused-vs-alternating comparison fail only when a PPOCore was created (and dropped) earlier in the process?
usage: dbg_after_core.py <mode>   mode: none | core | core_close | core_gc | torch_only"""
import sys, gc, numpy as np, torch
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
mode = sys.argv[1] if len(sys.argv) > 1 else "core"
dev = torch.device("cuda", 0)
if mode.startswith("core"):
    c = PPOCore(89, 90, (32, 32), (32, 32), use_bf16=False, max_rows=512)
    x = torch.randn(512, 89, device=dev); a = torch.zeros(512, dtype=torch.int32, device=dev); l = torch.zeros(512, device=dev)
    torch.cuda.synchronize(); c.act(x, a, l); c.sync()
    if mode == "core_close": c.close()
    if mode == "core_gc": del c; gc.collect()
elif mode == "torch_only":
    x = torch.randn(512, 89, device=dev); torch.cuda.synchronize()
T = 12; out = []
for fused in (False, True):
    cfg = _lib.default_gym_config(); cfg.no_touch_max_steps = 5; cfg.seed_lo = 31
    env = BatchedEnv(21, 2, cfg=cfg)
    N, D = env.n_agents, env.obs_size
    ppo = PPOCore(D, env.n_actions, (256, 256, 256), (64,), use_bf16=True, max_rows=max(N, 64), seed=5)
    obs = torch.zeros((T + 1, N, D), device=dev); acts = torch.zeros((T, N), dtype=torch.int32, device=dev)
    logp = torch.zeros((T, N), device=dev); rew = torch.zeros((T, N), device=dev); done = torch.zeros((T, N), dtype=torch.int32, device=dev)
    torch.cuda.synchronize(); env.reset(True, obs[0])
    if fused: assert env.collect(ppo, T, obs, acts, logp, rew, done)
    else:
        for t in range(T):
            ppo.act(obs[t], acts[t], logp[t]); ppo.sync(); env.step(acts[t], obs[t + 1], rew[t], done[t]); env.sync()
    env.sync(); torch.cuda.synchronize()
    out.append([x.cpu().numpy() for x in (obs, acts, logp, rew, done)])
for a, b, n in zip(out[0], out[1], ("obs", "acts", "logp", "rew", "done")):
    d = (a != b) if n != "logp" else (np.abs(a - b) > 1e-6)
    if d.any():
        idx = np.argwhere(d); print(mode, n, "differs:", len(idx), "first", idx[0], "agents", sorted(set(idx[:, 1].tolist()))[:12], "a", a[tuple(idx[0])], "b", b[tuple(idx[0])])
        if n == "obs": print("  cols at first t:", sorted(set(idx[idx[:, 0] == idx[0, 0]][:, 2].tolist()))[:40])
    else: print(mode, n, "equal")
