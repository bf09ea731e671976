"""Does it learn?  N iterations of the BASELINE configs[1] learner (1v1, 4096 envs, 262 144 agent-steps per iteration) from scratch;
prints the mean step reward, the policy entropy and the ball-touch rate along the way.  usage: train_probe.py [iterations] [epochs]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 4096 * 2 * 32
cfg = LearnerConfig(numEnvs=4096, teamSize=1, timestepsPerIteration=B, expBufferSize=B, randomSeed=1,
                    ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 4, epochs=epochs, policyLR=2e-4, criticLR=2e-4, entCoef=0.01, autocastLearn=True))
L = Learner(cfg)
t0 = time.time()
for it in range(iters):
    L.iteration()
    if it % max(1, iters // 12) == 0 or it == iters - 1:
        rep = L.finish_report()
        # a touch shows up as the TouchBall part of no reward term here; use the obs-independent signal: episodes that did not time out
        print("iter %4d  %6.1f M steps  %5.1f s  mean step reward %+.4f  entropy %.3f  value loss %.4f  done rate %.4f" % (
            it, L.total_timesteps / 1e6, time.time() - t0, float(L.rew_buf.mean().item()), rep["Policy Entropy"], rep["Value Function Loss"], float(L.done_buf.float().mean().item())))

import numpy as np
p = L.ppo.get_params(2)
print("soak: params finite", bool(np.isfinite(p).all()), "| obs finite", bool(torch.isfinite(L.obs_buf).all().item()), "| |obs| max %.2f" % float(L.obs_buf.abs().max().item()),
      "| reward range [%.2f, %.2f]" % (float(L.rew_buf.min().item()), float(L.rew_buf.max().item())), "| agent-steps/s overall %.2fM" % (L.total_timesteps / (time.time() - t0) / 1e6))

# how often the narrowphase queues overflowed into the inline fallback (release build: rlgpu_env_overflow_counts)
ovf = L.env.overflow_counts()
print("queue overflow events over the run: frontier", ovf[0], "ball region", ovf[1], "car region", ovf[2], "items", ovf[3], "pool", ovf[4],
      "(of %.0fM env-ticks)" % (L.total_timesteps / 2 * 8 / 1e6))
