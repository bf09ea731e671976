"""The HIP stepper against the host build of the same source on random control tapes (GPU box; no reference needed): kickoffs of
1v1 / 2v2 / 3v3 from the device's own KickoffState, every car on random controls held for random spans (everybody meets at the ball: car-ball,
car-car, wall contacts, heaps), both sides handed the same controls every tick and compared after EVERY tick for equality of every exchanged
field.  The env slots keep their broadphase history across the per-tick uploads (an upload is a SetState), the host side carries one per
arena (PortSim.step(hist=...)).          usage: random_tapes_hip.py [envs per team size] [ticks] [seed]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd.env import BatchedEnv, procedural_mesh
from rlgymppo_cpp_amd import _lib
from simlib import PortSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
v, t = procedural_mesh()
port = PortSim(); port.set_mesh(v, t)
total = 0
for team in (1, 2, 3):
    nc = 2 * team
    cfg = _lib.default_gym_config(); cfg.setter_kind = 1; cfg.seed_lo = seed + team
    env = BatchedEnv(n, team, cfg=cfg, mesh=(v, t))
    env.reset(True); env.sync()
    host = env.download_states()
    hists = [(C.c_uint16 * 8)() for _ in host]
    rng = np.random.RandomState(seed * 10 + team)
    tape = np.zeros((ticks, n, nc, 8), np.float32)
    for e in range(n):
        for k in range(nc):
            tt = 0
            while tt < ticks:
                span = int(rng.randint(4, 60)); c = np.zeros(8, np.float32)
                c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
                c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
                tape[tt:tt + span, e, k] = c; tt += span
    bad = None; touched = 0
    for tk in range(ticks):
        cur = env.download_states()
        for e in range(n):
            for k in range(nc):
                host[e].cars[k].controls[:] = list(tape[tk, e, k]); cur[e].cars[k].controls[:] = list(tape[tk, e, k])
        env.upload_states(cur); env.physics_ticks(1)
        cur = env.download_states()
        for e in range(n):
            port.step(host[e], 1, hist=hists[e])
            if bytes(host[e])[:C.sizeof(host[e]) - 0] != bytes(cur[e]) and bad is None:
                a = np.frombuffer(bytes(host[e]), np.uint8); b = np.frombuffer(bytes(cur[e]), np.uint8)
                bad = (tk + 1, e, int(np.argmax(a != b)))
        if bad: break
        total += n
    demo = sum(1 for e in range(n) for k in range(nc) if cur[e].cars[k].flags & (1 << 13))
    hit = sum(1 for e in range(n) for k in range(nc) if cur[e].cars[k].bh_tick_hit >= 0)
    print(f"{team}v{team}: {n} envs x {ticks} ticks: " + ("HIP and the host build EQUAL in every byte of the exchanged state after every tick" if bad is None else f"first difference after tick {bad[0]} in env {bad[1]} at byte {bad[2]}") + f"  (cars that have hit the ball: {hit}, demolished at the end: {demo})", flush=True)
    env.close()
print("env-ticks compared equal:", total)
