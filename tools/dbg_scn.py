import os, sys, numpy as np, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import ArenaState
from simlib import PortSim
sg = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
name = sys.argv[1] if len(sys.argv) > 1 else "ball_corner_fillets"
n_env = int(sys.argv[2]) if len(sys.argv) > 2 else 1
port = PortSim(); port.set_mesh(sg["mesh_verts"], sg["mesh_tris"])
env = BatchedEnv(n_env, 1, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
host = ArenaState.from_buffer_copy(sg[f"phys/{name}/start"].tobytes())
tape = sg[f"phys/{name}/tape"]
cur = [ArenaState.from_buffer_copy(bytes(host)) for _ in range(n_env)]
env.upload_states(cur)
for t in range(min(int(sys.argv[3]) if len(sys.argv) > 3 else 60, len(tape))):
    cur = env.download_states()
    for k in range(2):
        host.cars[k].controls[:] = list(tape[t, k])
        for c in cur: c.cars[k].controls[:] = list(tape[t, k])
    env.upload_states(cur); env.physics_ticks(1); cur = env.download_states()
    port.step(host, 1)
    d = max(abs(host.ball.pos[i] - cur[0].ball.pos[i]) for i in range(3)); dv = max(abs(host.ball.vel[i] - cur[0].ball.vel[i]) for i in range(3))
    print(t, "ball pos", [round(x, 3) for x in host.ball.pos], "dpos %.4g dvel %.4g" % (d, dv))
    if d > 1e-2 or dv > 1: break
