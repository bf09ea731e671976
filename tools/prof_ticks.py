"""Uniform-work microbenchmark: every env in the same state (cars at rest / driving), k_env_ticks timing via torch events."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import default_arena
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = BatchedEnv(n, 1)
for label, thr in (("rest", 0.0), ("throttle", 1.0)):
    s = default_arena(2)
    for k in range(2):
        s.cars[k].controls[0] = thr
    env.upload_states([s] * n)
    env.physics_ticks(8); env.sync()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        env.physics_ticks(8)
    e1.record(); torch.cuda.synchronize()
    print(label, "ms per 8 ticks:", e0.elapsed_time(e1) / 10)
