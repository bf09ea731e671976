"""The host build of the stepper against the LIVE reference (oracle/_ref) on random control tapes that no fixture holds: a kickoff of
1v1 / 2v2 / 3v3 (ResetToRandomKickoff), every car driven by random controls held for random spans -- everybody converges on the ball, so
car-ball, car-car and wall contacts, bumps and demolitions all occur -- compared in BULLET units after every tick, like
tools/raw_divergence.py.  With a fourth argument the live arena's car set is rehashed per tape (ref_arena_rehash), so the reference visits its
cars in a different order from tape to tape; with a fifth ("hunt") every car boosts at the nearest opponent, steering by the reference's
state of the tick before (demolitions, wrecks, respawns); with "walls" every car starts on a wall, beside or above a goal, or on the ceiling, at speed.\n          usage: random_tapes.py [tapes] [ticks] [first seed] [rehash|-] [hunt|walls|aerial|corners|scrum]"""
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
    if len(sys.argv) > 5 and sys.argv[5] == "hunt":      # hunt: full tanks -- a demolition needs a SUPERSONIC attacker (Arena.cpp:372-385), which a kickoff's 33 boost does not buy
        for k in range(nc): s0.cars[k].boost = 100.0
    if len(sys.argv) > 5 and sys.argv[5] == "walls":
        # wall play (round 6: the kickoff tapes live on the floor; a car beside a goal post on the back wall found the ray-leaf hole): every car starts
        # ON a side wall, a back wall next to / above the goal or the ceiling (clear of the fillets: a start INSIDE the geometry is not a state the game reaches),
        # wheels on the surface, at speed, tank full
        for k in range(nc):
            kind = rng.randint(5)
            if kind == 0:   side = rng.choice([-1.0, 1.0]); up = np.array([-side, 0, 0]); pos = np.array([side * (4096 - 17.0), rng.uniform(-3400, 3400), rng.uniform(420, 1600)])
            elif kind == 1: side = rng.choice([-1.0, 1.0]); up = np.array([0, -side, 0]); pos = np.array([rng.choice([-1.0, 1.0]) * rng.uniform(1100, 2800), side * (5120 - 17.0), rng.uniform(420, 1600)])
            elif kind == 2: side = rng.choice([-1.0, 1.0]); up = np.array([0, -side, 0]); pos = np.array([rng.choice([-1.0, 1.0]) * rng.uniform(960, 1100), side * (5120 - 17.0), rng.uniform(420, 800)])   # beside a goal post
            elif kind == 3: up = np.array([0, 0, -1.0]); pos = np.array([rng.uniform(-3000, 3000), rng.uniform(-4000, 4000), 2044 - 17.0])
            else:           side = rng.choice([-1.0, 1.0]); up = np.array([0, -side, 0]); pos = np.array([rng.uniform(-800, 800), side * (5120 - 17.0), rng.uniform(760, 1600)])   # above the goal mouth
            t1 = np.cross(up, [0.3, 0.5, 0.8]); t1 /= np.linalg.norm(t1); ang = rng.uniform(0, 2 * np.pi)
            fwd = np.cos(ang) * t1 + np.sin(ang) * np.cross(up, t1); right = np.cross(up, fwd)
            c = s0.cars[k]
            c.pos[:] = [float(x) for x in pos]; c.rot[:] = [float(x) for x in np.concatenate([fwd, right, up])]
            c.vel[:] = [float(x) for x in fwd * rng.uniform(300, 2200)]; c.ang_vel[:] = [0.0, 0.0, 0.0]; c.boost = 100.0
        s0.ball.pos[:] = [float(rng.uniform(-3000, 3000)), float(rng.uniform(-4000, 4000)), float(rng.uniform(100, 1800))]
        s0.ball.vel[:] = [float(x) for x in rng.uniform(-1500, 1500, 3)]
    if len(sys.argv) > 5 and sys.argv[5] == "scrum":
        # every car on the floor in a ring around the ball, facing it, at speed: a pile-up in the first second -- car-car boxes, deep contacts, bumps, several cars
        # on the ball at once, wheels on roofs
        cx, cy = rng.uniform(-2500, 2500), rng.uniform(-3500, 3500)
        s0.ball.pos[:] = [float(cx), float(cy), 93.15]; s0.ball.vel[:] = [0.0, 0.0, 0.0]
        ang0 = rng.uniform(0, 2 * np.pi)
        for k in range(nc):
            a_k = ang0 + 2 * np.pi * k / nc + rng.uniform(-0.25, 0.25); rad = rng.uniform(700, 1300)
            pos = np.array([cx + rad * np.cos(a_k), cy + rad * np.sin(a_k), 17.0]); yaw = a_k + np.pi + rng.uniform(-0.15, 0.15)
            c = s0.cars[k]
            c.pos[:] = [float(x) for x in pos]; c.rot[:] = [float(np.cos(yaw)), float(np.sin(yaw)), 0.0, float(-np.sin(yaw)), float(np.cos(yaw)), 0.0, 0.0, 0.0, 1.0]
            v = rng.uniform(1200, 2250); c.vel[:] = [float(np.cos(yaw) * v), float(np.sin(yaw) * v), 0.0]; c.ang_vel[:] = [0.0, 0.0, 0.0]; c.boost = 100.0
    if len(sys.argv) > 5 and sys.argv[5] in ("aerial", "corners"):
        # aerial: every car in the air around the ball, any orientation, spinning, tank full -- air control, flips, ball hits in the air, landings on whatever comes
        # corners: every car on the floor in a corner region or in front of a goal, heading for the 45-degree wall / the fillets / the goal frame at speed
        def rand_rot():
            q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
            return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        bp = np.array([rng.uniform(-2500, 2500), rng.uniform(-3500, 3500), rng.uniform(400, 1500)]) if sys.argv[5] == "aerial" else np.array([rng.uniform(-3000, 3000), rng.uniform(-4000, 4000), 93.15])
        s0.ball.pos[:] = [float(x) for x in bp]; s0.ball.vel[:] = [float(x) for x in rng.uniform(-900, 900, 3) * (1.0 if sys.argv[5] == "aerial" else np.array([1, 1, 0]))]
        for k in range(nc):
            c = s0.cars[k]
            if sys.argv[5] == "aerial":
                for _ in range(200):      # clear of the other cars (a start with two hitboxes inside one another is not a state the game reaches)
                    R = rand_rot(); off = rng.normal(size=3); off *= rng.uniform(250, 900) / np.linalg.norm(off); pos = bp + off; pos[2] = min(max(pos[2], 150.0), 1750.0)
                    if all(np.linalg.norm(pos - np.array(list(s0.cars[j].pos))) > 330.0 for j in range(k)): break
                c.pos[:] = [float(x) for x in pos]; c.rot[:] = [float(x) for x in np.concatenate([R[:, 0], R[:, 1], R[:, 2]])]
                c.vel[:] = [float(x) for x in (bp - pos) / np.linalg.norm(bp - pos) * rng.uniform(200, 1800) + rng.uniform(-200, 200, 3)]
                c.ang_vel[:] = [float(x) for x in rng.uniform(-4.5, 4.5, 3)]; c.flags = c.flags & ~0x1f        # off the ground
            else:
                sx, sy = rng.choice([-1.0, 1.0]), rng.choice([-1.0, 1.0])
                if rng.rand() < 0.6: pos = np.array([sx * rng.uniform(2300, 3300), sy * rng.uniform(3300, 4300), 17.0]); aim = np.array([sx * 4096, sy * 5120, 17.0])      # towards a corner
                else:                pos = np.array([rng.uniform(-1100, 1100), sy * rng.uniform(3600, 4600), 17.0]); aim = np.array([rng.choice([-1.0, 1.0]) * rng.uniform(700, 1100), sy * 5200, 17.0])   # towards a goal post
                d = aim - pos; yaw = float(np.arctan2(d[1], d[0])) + rng.uniform(-0.4, 0.4)
                c.pos[:] = [float(x) for x in pos]; c.rot[:] = [float(np.cos(yaw)), float(np.sin(yaw)), 0.0, float(-np.sin(yaw)), float(np.cos(yaw)), 0.0, 0.0, 0.0, 1.0]
                c.vel[:] = [float(np.cos(yaw) * v) for v in [rng.uniform(800, 2200)]] + [0.0, 0.0]; c.vel[1] = float(np.sin(yaw) * np.hypot(c.vel[0], 0) / max(abs(np.cos(yaw)), 1e-3)) if False else float(np.sin(yaw) * abs(c.vel[0]) / max(abs(np.cos(yaw)), 0.2))
                c.ang_vel[:] = [0.0, 0.0, 0.0]
            c.boost = 100.0
    live = None
    if os.environ.get("RT_LIVE"):
        # RT_LIVE=<file>:<case>: start state and actions of one rollout recorded by tools/live_gym_hip.py --record, as a PHYSICS tape (the action table's controls held for
        # the tick skip) -- with RT_DEBUG=1 the tick, body and contact lists where a Gym rollout that differs leaves the reference
        path, case = os.environ["RT_LIVE"].rsplit(":", 1); live = np.load(path)
        s0 = ArenaState.from_buffer_copy(live[f"gym/{case}/start_raw"].tobytes()); team = int(live[f"gym/{case}/cfg"][0]); nc = 2 * team
        ref.lib.ref_arena_free(a); a = ref.arena(team)
        if s0.mutators.flags & ~32: os.environ["RT_MUT"] = "M1"
    if len(sys.argv) > 4 and sys.argv[4] != "-": ref.lib.ref_arena_rehash(a, 1 + (seed * 7) % 60)
    if os.environ.get("RT_MUT"):        # RT_MUT=M1 | M2: the arena under one of tests/golden/make_mutator_golden.py's non-default MutatorConfigs (both sides)
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); from make_mutator_golden import mutator_sets
        _, mm, dd = [x for x in mutator_sets() if x[0] == os.environ["RT_MUT"]][0]
        ref.lib.ref_arena_set_mutators.argtypes = [C.c_void_p, C.c_void_p, C.c_float]; ref.lib.ref_arena_set_mutators(a, C.byref(mm), C.c_float(dd))
    ref.set_state(a, s0); got0 = ref.get_state(a); s0.car_order = got0.car_order
    if os.environ.get("RT_MUT"): s0.mutators = got0.mutators; s0.hidden.valid |= 8
    # both sides draw the respawn slots from the same engine state (RlgpuArenaHidden::ref_engine; oracle/ref_driver.cpp:ref_seed_engine): a tape stays
    # comparable through its respawns
    engine0 = 1 + (seed * 2654435761) % 2147483645 if live is None else int(s0.hidden.ref_engine)
    ref.lib.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= 4; s0.hidden.ref_engine = engine0
    # parity mode for the wheel rays too (RLGPU_MUT_RAY_PROXY_LISTS, csrc/arena_world.h: every dynamic body on the broadphase's list gets the convex cast, as in the
    # reference); RT_NO_RAY_LISTS=1 shows what the product's default -- the box test alone -- leaves out
    if not os.environ.get("RT_NO_RAY_LISTS"): s0.hidden.valid |= 8; s0.mutators.flags |= 32
    tape = np.zeros((ticks, nc, 8), np.float32)
    for k in range(nc):
        t = 0
        while t < ticks:
            span = int(rng.randint(4, 60))
            c = np.zeros(8, np.float32)
            c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
            c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
            tape[t:t + span, k] = c; t += span
    if live is not None:
        tab = np.zeros((128, 8), np.float32); ref.lib.ref_action_table(tab.ctypes.data_as(C.c_void_p), 128)
        acts = live[f"gym/{case}/actions"]; skip = int(live[f"gym/{case}/cfg"][1]); ticks = len(acts) * skip
        tape = np.ascontiguousarray(np.repeat(tab[acts], skip, axis=0), np.float32)
    hunt = len(sys.argv) > 5 and sys.argv[5] == "hunt"
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
    if first is not None and os.environ.get("RT_DEBUG"):
        # where the two part: the first differing body, its wheels and the contact lists of that tick on both sides (re-run from the start: the reference's arena has moved on)
        t = first - 1; d = np.argwhere(bp[t] != br[t]); body = int(d[0][0])
        print("   differing (body, field):", d[:8].tolist()); print("   port", raw_p[t][body].tolist()); print("   ref ", raw_r[t][body].tolist())
        b = ref.arena(team)
        if len(sys.argv) > 4 and sys.argv[4] != "-": ref.lib.ref_arena_rehash(b, 1 + (seed * 7) % 60)
        if os.environ.get("RT_MUT"): ref.lib.ref_arena_set_mutators(b, C.byref(mm), C.c_float(dd))
        ref.set_state(b, s0); ref.lib.ref_seed_engine(C.c_uint32(engine0))
        for tt in range(first):
            for k in range(nc): ref.set_controls(b, k, tape[tt, k])
            ref.step(b, 1)
        if body >= 1:
            w = np.zeros((4, 12), np.float32); ref.lib.ref_debug_wheels(b, body - 1, w.ctypes.data_as(C.c_void_p)); print("   reference wheels of car", body - 1); print(w.round(4))
        if body >= 1:
            port.lib.port_run_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            s4 = ArenaState.from_buffer_copy(bytes(s0)); o4 = (ArenaState * 1)()
            if first > 1: port.lib.port_run_tape(C.byref(s4), tape.ctypes.data, first - 1, first - 1, C.byref(o4)); s4 = o4[0]
            for k in range(nc): s4.cars[k].controls[:] = [float(x) for x in tape[first - 1, k]]
            w2 = np.zeros((4, 12), np.float32); port.lib.port_debug_wheels(C.byref(s4), body - 1, w2.ctypes.data_as(C.c_void_p)); print("   port wheels (from its own state one tick earlier, handed over in uu)"); print(w2.round(4))
        buf = np.zeros((64, 16), np.float32); n = ref.lib.ref_debug_manifolds(b, buf.ctypes.data_as(C.c_void_p), 64)
        print("   reference manifold points (body0, body1, manifold, lifetime | normal | distance, applied):")
        for i in range(n): print("    ", buf[i, :4], buf[i, 10:13].round(4), buf[i, 13], buf[i, 14])
        io = np.zeros((64, 2), np.float32); ni = ref.lib.ref_debug_island_order(b, io.ctypes.data_as(C.c_void_p), 64)
        print("   reference: the solver's manifold order (body0, body1):", [tuple(int(x) for x in io[i]) for i in range(ni)])
        out = np.zeros((64, 16), np.float32); port.lib.port_run_tape_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        s5 = ArenaState.from_buffer_copy(bytes(s0)); n2 = port.lib.port_run_tape_contacts(C.byref(s5), tape.ctypes.data, first, out.ctypes.data_as(C.c_void_p), 64, None, None)
        print("   port contacts of that tick (a, b, sid, special | normal | distance, applied):")
        for i in range(n2): print("    ", out[i, :4], out[i, 10:13].round(4), out[i, 13], out[i, 14])
        ref.lib.ref_arena_free(b)
    ref.lib.ref_arena_free(a)
print(f"{exact} of {n_tapes} random tapes bit-identical to the live reference over {ticks} ticks; {exact_ticks} ticks compared equal in all")
