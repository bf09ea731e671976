"""Scratch diagnostics run on the GPU box (not part of the test suite)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd import _lib
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import default_arena
from simlib import PortSim, port_gym_cfg, port_gym_reset, port_gym_step

n = 64
env = BatchedEnv(n, 1)
obs = env.reset(True); env.sync()
port = PortSim(); v, t = port.procedural_mesh(); port.set_mesh(v, t)
hs, hobs = port_gym_reset(port, [default_arena(2) for _ in range(n)], port_gym_cfg(), run_setter=True)
o = obs.cpu().numpy()
d = np.abs(o - hobs)
print("reset obs max diff", d.max(), "at", np.unravel_index(d.argmax(), d.shape))
print("gpu row0", o[0, :12], "\nhost row0", hobs[0, :12])
st = env.download_states(2)
print("gpu ball", list(st[0].ball.pos), "host ball", list(hs[0].ball.pos))
print("gpu car0 pos", list(st[0].cars[0].pos), "host", list(hs[0].cars[0].pos), "reset_count", st[0].gym.reset_count, hs[0].gym.reset_count)
