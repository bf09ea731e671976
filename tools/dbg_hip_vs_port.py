"""Tick-by-tick HIP vs host port on one golden physics scenario, re-synced to the port's state every tick: prints every tick whose one-tick
results differ by more than rounding (development tool; GPU box).   python tools/dbg_hip_vs_port.py <scenario> [first_tick last_tick]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd.env import BatchedEnv
from rlgymppo_cpp_amd.state import ArenaState
from simlib import PortSim
sg = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
name = sys.argv[1]; t0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0; t1 = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
port = PortSim(); port.set_mesh(sg["mesh_verts"], sg["mesh_tris"])
host = ArenaState.from_buffer_copy(sg[f"phys/{name}/start"].tobytes())
nc = host.num_cars
env = BatchedEnv(1, nc // 2, mesh=(sg["mesh_verts"], sg["mesh_tris"]))
tape = sg[f"phys/{name}/tape"]
for t in range(min(len(tape), t1)):
    for k in range(nc): host.cars[k].controls[:] = list(tape[t, k])
    env.upload_states([host]); env.physics_ticks(1); cur = env.download_states()[0]
    port.step(host, 1)
    if t < t0: continue
    bad = []
    for k in range(nc):
        a, b = host.cars[k], cur.cars[k]
        dv = max(abs(x - y) for x, y in zip(a.vel, b.vel)); dw = max(abs(x - y) for x, y in zip(a.ang_vel, b.ang_vel)); dp = max(abs(x - y) for x, y in zip(a.pos, b.pos))
        if dv > 1e-2 or dw > 1e-3 or dp > 1e-2 or a.flags != b.flags: bad.append(f"car{k} dpos {dp:.3g} dvel {dv:.3g} dang {dw:.3g} flags {a.flags:#x}/{b.flags:#x}")
    dv = max(abs(x - y) for x, y in zip(host.ball.vel, cur.ball.vel))
    if dv > 1e-2: bad.append(f"ball dvel {dv:.3g}")
    if bad: print(t, "; ".join(bad))
print("done")
