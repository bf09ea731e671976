"""One rank of a two-rank run of the PYTHON host on one GPU (tests/test_host_cpp.py::test_python_host_two_ranks_on_one_gpu): the launcher's
environment names the ranks, RLGPU_COMM_TRANSPORT=shm carries the collectives (RCCL refuses two ranks on one device).  Prints a checksum of
the parameters after <iterations> iterations."""
import os, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from rlgymppo_cpp_amd.learner import Learner, LearnerConfig, PPOLearnerConfig
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
ne = 128; B = ne * 2 * 8
L = Learner(LearnerConfig(numEnvs=ne, teamSize=1, timestepsPerIteration=B, expBufferSize=B, randomSeed=5,
                          ppo=PPOLearnerConfig(batchSize=B, miniBatchSize=B // 2, epochs=1, policyLR=2e-4, criticLR=2e-4, entCoef=0.01, autocastLearn=True)))
for _ in range(iters):
    L.iteration()
torch.cuda.synchronize()
p = L.ppo.get_params(2)
print(f"rank {L.rank} of {L.world}: device {L.cfg.device}, parameter checksum {hashlib.sha256(p.tobytes()).hexdigest()[:16]}", flush=True)
L.comm.close() if hasattr(L.comm, "close") else None
