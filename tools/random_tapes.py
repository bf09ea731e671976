"""The host build of the stepper against the LIVE reference (oracle/_ref) on random control tapes that no fixture holds: a kickoff of
1v1 / 2v2 / 3v3 (ResetToRandomKickoff), every car driven by random controls held for random spans -- everybody converges on the ball, so
car-ball, car-car and wall contacts, bumps and demolitions all occur -- compared in BULLET units after every tick, like
tools/raw_divergence.py.  With a fourth argument the live arena's car set is rehashed per tape (ref_arena_rehash), so the reference visits its
cars in a different order from tape to tape; with a fifth ("hunt") every car boosts at the nearest opponent, steering by the reference's
state of the tick before (demolitions, wrecks, respawns).          usage: random_tapes.py [tapes] [ticks] [first seed] [rehash|-] [hunt]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simlib import PortSim, RefSim
from rlgymppo_cpp_amd.state import ArenaState
n_tapes = int(sys.argv[1]) if len(sys.argv) > 1 else 30
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 600
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
gold = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
verts, tris = gold["mesh_verts"], gold["mesh_tris"]
port = PortSim(); port.set_mesh(verts, tris); ref = RefSim(verts, tris)
port.lib.port_run_tape_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
ref.lib.ref_arena_get_raw.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
ref.lib.ref_arena_reset_kickoff.argtypes = [C.c_void_p, C.c_int]
ref.lib.ref_arena_free.argtypes = [C.c_void_p]; ref.lib.ref_arena_rehash.argtypes = [C.c_void_p, C.c_int]
exact = 0; exact_ticks = 0
for seed in range(seed0, seed0 + n_tapes):
    rng = np.random.RandomState(seed)
    team = 1 + seed % 3; nc = 2 * team
    k0 = ref.arena(team); ref.lib.ref_arena_reset_kickoff(k0, seed); s0 = ref.get_state(k0); ref.lib.ref_arena_free(k0)
    a = ref.arena(team)
    if len(sys.argv) > 5:      # hunt: full tanks -- a demolition needs a SUPERSONIC attacker (Arena.cpp:372-385), which a kickoff's 33 boost does not buy
        for k in range(nc): s0.cars[k].boost = 100.0
    if len(sys.argv) > 4 and sys.argv[4] != "-": ref.lib.ref_arena_rehash(a, 1 + (seed * 7) % 60)
    ref.set_state(a, s0); s0.car_order = ref.get_state(a).car_order
    # both sides draw the respawn slots from the same engine state (RlgpuArenaHidden::ref_engine; oracle/ref_driver.cpp:ref_seed_engine): a tape stays
    # comparable through its respawns
    engine0 = 1 + (seed * 2654435761) % 2147483645
    ref.lib.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= 4; s0.hidden.ref_engine = engine0
    tape = np.zeros((ticks, nc, 8), np.float32)
    for k in range(nc):
        t = 0
        while t < ticks:
            span = int(rng.randint(4, 60))
            c = np.zeros(8, np.float32)
            c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
            c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
            tape[t:t + span, k] = c; t += span
    hunt = len(sys.argv) > 5
    raw_r = np.zeros((ticks, 1 + nc, 18), np.float32)
    n_demo = 0; n_resp = 0; was = [False] * nc
    for t in range(ticks):
        if hunt:      # the tape is written as the reference runs: full throttle and boost at the nearest opponent
            cur = ref.get_state(a)
            for k in range(nc):
                me = cur.cars[k]; opp = [cur.cars[j] for j in range(nc) if j % 2 != k % 2 and not (cur.cars[j].flags & (1 << 13))]
                c = np.zeros(8, np.float32); c[0] = 1.0; c[6] = 1.0
                if opp:
                    o = min(opp, key=lambda q: (q.pos[0] - me.pos[0]) ** 2 + (q.pos[1] - me.pos[1]) ** 2)
                    dx, dy = o.pos[0] - me.pos[0], o.pos[1] - me.pos[1]
                    fx, fy = me.rot[0], me.rot[1]          # forward axis = the first three floats of rot[9] (RlgpuCarState: forward, right, up)
                    cross = fx * dy - fy * dx; along = fx * dx + fy * dy
                    ang = float(np.arctan2(cross, along))          # the opponent's bearing, left positive
                    c[1] = float(np.clip(-2.0 * ang, -1.0, 1.0))    # steer towards it (a bang-bang steer never lines the bumper up)
                    c[6] = 1.0 if abs(ang) < 0.6 else 0.0           # boost when it is ahead
                    c[7] = 1.0 if abs(ang) > 1.5 else 0.0           # powerslide around when it is behind
                tape[t, k] = c
                dm = bool(me.flags & (1 << 13)); n_demo += dm and not was[k]; n_resp += was[k] and not dm; was[k] = dm
        for k in range(nc):
            ref.set_controls(a, k, tape[t, k])
        ref.step(a, 1)
        ref.lib.ref_arena_get_raw(a, nc, raw_r[t].ctypes.data)
    raw_p = np.zeros((ticks, 1 + nc, 18), np.float32)
    st = ArenaState.from_buffer_copy(bytes(s0))
    port.lib.port_run_tape_raw(C.byref(st), tape.ctypes.data, ticks, raw_p.ctypes.data)
    fin = ref.get_state(a)
    demos = sum(1 for k in range(nc) if fin.cars[k].flags & (1 << 13))
    if hunt: print(f"   demolitions during the tape: {n_demo}, respawns: {n_resp}")
    bp, br = raw_p.view(np.uint32), raw_r.view(np.uint32)
    first = next((t + 1 for t in range(ticks) if (bp[t] != br[t]).any()), None)
    exact += first is None; exact_ticks += (ticks if first is None else first - 1)
    print(f"seed {seed} {team}v{team} order {s0.car_order:x}: " + ("bit-identical for all %d ticks" % ticks if first is None else "first raw difference after tick %d" % first) + (f"  ({demos} car(s) demolished at the end)" if demos else ""), flush=True)
    ref.lib.ref_arena_free(a)
print(f"{exact} of {n_tapes} random tapes bit-identical to the live reference over {ticks} ticks; {exact_ticks} ticks compared equal in all")
