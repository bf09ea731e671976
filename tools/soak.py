"""Two soaks on the round's final build (last row):
  A  identical collection flows at BASELINE configs[1] (1v1, 4096 envs): two env batches, one learner with the sampler rewound, <launches> fused
     collection launches of 12 gym steps each -- outputs and downloaded resident states compared byte for byte after every launch;
  B  a learner that LEARNS (Python host, 2 epochs per iteration, 1v1 / 2048 envs, tessellated 16-object arena) for <iterations> iterations with
     RLGPU_REDZONE guard bytes behind every device buffer of the env batch and of the learner, checked at the end, with the counters of the
     narrowphase fallbacks and of lost contact points.
usage: soak.py [launches] [iterations] [team size of B] [envs of B]      (launches = 0: B only)"""
import os, sys, time
os.environ["RLGPU_REDZONE"] = "65536"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.ppo import PPOCore
from rlgymppo_cpp_amd import _lib
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 60
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
team_b = int(sys.argv[3]) if len(sys.argv) > 3 else 1
envs_b = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
dev = torch.device("cuda", 0); CAP = 12
n = 4096
def soak_a():
    ea, eb = BatchedEnv(n, 1), BatchedEnv(n, 1)
    core = PPOCore(ea.obs_size, ea.n_actions, (256, 256, 256), (256, 256, 256), use_bf16=True, max_rows=ea.n_agents)
    N, D = ea.n_agents, ea.obs_size
    def bufs():
        return (torch.zeros((CAP + 1, N, D), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev), torch.zeros((CAP, N), device=dev),
                torch.zeros((CAP, N), device=dev), torch.zeros((CAP, N), dtype=torch.int32, device=dev))
    A, B = bufs(), bufs(); torch.cuda.synchronize()
    ea.reset(True, A[0][0]); eb.reset(True, B[0][0]); ea.sync(); eb.sync()
    t0 = time.time(); bad_at = None
    for k in range(launches):
        s, c = core.get_sampler()
        assert ea.collect(core, CAP, *A); ea.sync()
        core.set_sampler(s, c)
        assert eb.collect(core, CAP, *B); eb.sync()
        same = all(torch.equal(x, y) for x, y in zip(A, B))
        if same and (k % 10 == 9 or k == launches - 1):
            sa, sb = ea.download_states(), eb.download_states()
            same = all(bytes(sa[e]) == bytes(sb[e]) for e in range(n))
        if not same: bad_at = k; break
        A[0][0].copy_(A[0][CAP]); B[0][0].copy_(B[0][CAP]); torch.cuda.synchronize()
    ea.check_redzones(); eb.check_redzones(); core.check_redzones()
    print(f"A: {launches} launches x {CAP} gym steps x {n} envs x 2 flows ({launches * CAP * n * 8 * 2 / 1e6:.0f} M env-ticks): " + ("flows identical, outputs every launch and states every 10th" if bad_at is None else f"FLOWS PART at launch {bad_at}") + f"; redzones clean ({time.time() - t0:.0f} s)")
    ea.close(); eb.close(); core.close()

if launches > 0: soak_a()

import bench
from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
mesh = os.path.join(bench.make_tessellated_mesh_dir()[0], "soccar")
ne = envs_b; Bsz = ne * 2 * team_b * 32
L = Learner(LearnerConfig(numEnvs=ne, teamSize=team_b, timestepsPerIteration=Bsz, expBufferSize=Bsz, randomSeed=1,
                          ppo=PPOLearnerConfig(batchSize=Bsz, miniBatchSize=Bsz // 4, epochs=2, policyLR=2e-4, criticLR=2e-4, entCoef=0.01, autocastLearn=True)), mesh=mesh)
L.env.overflow_counts(reset=True); L.env.lost_contact_count(reset=True); L.env.epa_counts(reset=True); L.env.big_layout_ticks(reset=True)
t0 = time.time(); rews = []
for i in range(iters):
    L.iteration()
    if i % 50 == 49 or i == iters - 1:
        torch.cuda.synchronize(); rews.append(round(float(L.rew_buf.mean().item()), 4))
L.env.check_redzones(); L.ppo.check_redzones()
ticks = iters * 32 * ne * 8
print(f"B: {team_b}v{team_b}, {ne} envs, {iters} learning iterations ({ticks / 1e6:.0f} M env-ticks, tessellated arena), mean step reward every 50 iterations {rews}: redzones clean; "
      f"lost-contact events {L.env.lost_contact_count()}, env-ticks redone with the big contact layout {L.env.big_layout_ticks()}, exact fallbacks {L.env.overflow_counts()}, EPA queries {L.env.epa_counts()} ({time.time() - t0:.0f} s)")
