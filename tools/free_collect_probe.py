"""Free-running collection (rlgpu_collect_free) next to the lockstep launch (rlgpu_collect) on one box:
  * every env's rows under the free-running launch equal the lockstep launch's first steps[env] rows, bit for bit;
  * launch time of both, after `warm` lockstep launches that age the episodes.
usage: python tools/free_collect_probe.py [n_envs] [T] [warm] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
team = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda", 0)
env = BatchedEnv(n_envs, team)
core = PPOCore(env.obs_size, env.n_actions, use_bf16=True, max_rows=65536)
N, D = env.n_agents, env.obs_size
CAP = 2 * T
def bufs():
    return (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
            torch.zeros((CAP, N), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
obs, act, logp, rew, done = bufs()
obs2, act2, logp2, rew2, done2 = bufs()
steps = torch.zeros(n_envs, dtype=torch.int32, device=dev)
# two identical env batches (same seed) taken through the same warm-up with the same sampler counters: the lockstep launch of CAP steps on the
# first, the free-running launch on the second
env_b = BatchedEnv(n_envs, team)
s_stream, s_ctr0 = core.get_sampler()
env.reset(True, obs[0])
for i in range(warm):
    assert env.collect(core, T, obs, act, logp, rew, done)
    obs[0].copy_(obs[T])
assert env.collect(core, CAP, obs, act, logp, rew, done); env.sync()
ref = [x.clone() for x in (obs, act, logp, rew, done)]
core.set_sampler(s_stream, s_ctr0)
env_b.reset(True, obs2[0])
for i in range(warm):
    assert env_b.collect(core, T, obs2, act2, logp2, rew2, done2)
    obs2[0].copy_(obs2[T])
ok = env_b.collect_free(core, CAP, T * N, obs2, act2, logp2, rew2, done2, steps)
assert ok, "collect_free refused"
env_b.sync()
ms_free = env_b.last_step_ms()
st = steps.cpu().numpy()
P = N // n_envs
print("free launch: %.3f ms, steps per env min %d mean %.2f max %d, agent-steps %d (target %d)" % (ms_free, st.min(), st.mean(), st.max(), st.sum() * P, T * N))
bad = 0
for name, a, b in (("act", ref[1], act2), ("logp", ref[2], logp2), ("rew", ref[3], rew2), ("done", ref[4], done2)):
    a = a.cpu().numpy().reshape(CAP, n_envs, P); b = b.cpu().numpy().reshape(CAP, n_envs, P)
    mask = (np.arange(CAP)[:, None] < st[None, :])[:, :, None]
    diff = ((a != b) & mask).any(axis=(0, 2))
    print("  %-5s envs with a differing row: %d" % (name, diff.sum())); bad += diff.sum()
o1 = ref[0].cpu().numpy().reshape(CAP + 1, n_envs, P * D); o2 = obs2.cpu().numpy().reshape(CAP + 1, n_envs, P * D)
mask = (np.arange(CAP + 1)[:, None] <= st[None, :])[:, :, None]
diff = ((o1 != o2) & mask).any(axis=(0, 2)); print("  obs   envs with a differing row: %d" % diff.sum()); bad += diff.sum()
print("PREFIX-EQUAL" if bad == 0 else "PREFIX DIFFERS in %d env-arrays" % bad)
# timing: alternate lockstep T and free target T*N from the running state
def carry(o, s=None):
    if s is None: o[0].copy_(o[T]); return
    idx = s.long().repeat_interleave(P)
    o[0].copy_(o[idx, torch.arange(N, device=dev)])
tl, tf, ssum = [], [], []
for r in range(reps):
    assert env.collect(core, T, obs, act, logp, rew, done); env.sync(); tl.append(env.last_step_ms()); carry(obs)
    assert env.collect_free(core, CAP, T * N, obs, act, logp, rew, done, steps); env.sync(); tf.append(env.last_step_ms()); carry(obs, steps); ssum.append(int(steps.sum()) * P)
print("lockstep T=%d: %s ms (mean %.3f)" % (T, " ".join("%.2f" % x for x in tl), np.mean(tl)))
print("free target=%d: %s ms (mean %.3f), agent-steps %s" % (T * N, " ".join("%.2f" % x for x in tf), np.mean(tf), ssum))
print("per agent-step: lockstep %.4f us, free %.4f us" % (np.mean(tl) * 1e3 / (T * N), np.sum(tf) * 1e3 / np.sum(ssum)))
