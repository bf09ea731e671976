#!/usr/bin/env python3
"""Stress tapes of the live reference that carry round-6 finds; another one, `M2/cannon`, is in mutator_golden.npz.

  walls_9044   tools/random_tapes.py's `walls` mode as it was when the tape was found (the generator below keeps those start ranges), seed 9044, 3v3, 400 ticks: at
               tick 360 a wheel ray's END point lies on the side wall's plane to the last bit.  The analytic sign test drops such a ray; the two triangles
               btStaticPlaneShape::processAllTriangles spans -- their normal and offset come out of rounded vertices -- report a hit at fraction ~1, and Bullet
               takes it (btStaticPlaneShape.cpp:56-82, btRaycastCallback.cpp:34-60).  csrc/arena_world.h ray_planes leaves the boundary to the triangles now; built
               with -DRLG_TEST_ANALYTIC_PLANE_SIGN the stepper leaves this tape at tick 360 again (checked when the fixture was made).

  aerial_30118 / _80656 / _80142   tools/random_tapes.py's `aerial` mode (frozen below), 2v2 / 2v2 / 1v1, 300 ticks: cars tumbling around the ball; at tick 46 (83, 35) a wheel ray gets a
               convex-cast "hit" on a car it passes 20 - 30 uu away from (see main()).  Pinned with RLGPU_MUT_RAY_PROXY_LISTS set in the start state.

  aerial_80921   the same mode, 3v3: at tick 29 FOUR cars touch the ball at once.  Every touch adds its extra hit velocity to the ball's impulse cache
               (Arena::_BtCallback_OnCarBallCollision), in the order the broadphase made the ball's pairs -- the cars' arrival ranks in its cell's list, not their
               slots -- and from three terms on the order shows in the sum's last bit (csrc/arena_step.h collide_merge).

Layout as respawn_golden.npz: start state (engine in its hidden block), controls per tick, the reference's state and engine every 10 ticks.
usage: python tests/golden/make_edge_golden.py      (needs oracle/_ref: make -C oracle ref)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from simlib import RefSim, state_vec  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def wall_case(ref, seed, ticks):
    """tools/random_tapes.py `walls` (round 6, first version of its start ranges): (team, start state, engine, rehash, tape)"""
    rng = np.random.RandomState(seed)
    team = 1 + seed % 3; nc = 2 * team
    k0 = ref.arena(team); ref.lib.ref_arena_reset_kickoff(k0, seed); s0 = ref.get_state(k0); ref.lib.ref_arena_free(k0)
    for k in range(nc):
        kind = rng.randint(5)
        if kind == 0:   side = rng.choice([-1.0, 1.0]); up = np.array([-side, 0, 0]); pos = np.array([side * (4096 - 17.0), rng.uniform(-4200, 4200), rng.uniform(200, 1800)])
        elif kind == 1: side = rng.choice([-1.0, 1.0]); up = np.array([0, -side, 0]); pos = np.array([rng.uniform(-3200, 3200), side * (5120 - 17.0), rng.uniform(200, 1800)])
        elif kind == 2: side = rng.choice([-1.0, 1.0]); up = np.array([0, -side, 0]); pos = np.array([rng.choice([-1.0, 1.0]) * rng.uniform(893, 1100), side * (5120 - 17.0), rng.uniform(100, 800)])
        elif kind == 3: up = np.array([0, 0, -1.0]); pos = np.array([rng.uniform(-3400, 3400), rng.uniform(-4400, 4400), 2044 - 17.0])
        else:           side = rng.choice([-1.0, 1.0]); up = np.array([0, -side, 0]); pos = np.array([rng.uniform(-880, 880), side * (5120 - 17.0), rng.uniform(660, 1800)])
        t1 = np.cross(up, [0.3, 0.5, 0.8]); t1 /= np.linalg.norm(t1); ang = rng.uniform(0, 2 * np.pi)
        fwd = np.cos(ang) * t1 + np.sin(ang) * np.cross(up, t1); right = np.cross(up, fwd)
        c = s0.cars[k]
        c.pos[:] = [float(x) for x in pos]; c.rot[:] = [float(x) for x in np.concatenate([fwd, right, up])]
        c.vel[:] = [float(x) for x in fwd * rng.uniform(300, 2200)]; c.ang_vel[:] = [0.0, 0.0, 0.0]; c.boost = 100.0
    s0.ball.pos[:] = [float(rng.uniform(-3000, 3000)), float(rng.uniform(-4000, 4000)), float(rng.uniform(100, 1800))]
    s0.ball.vel[:] = [float(x) for x in rng.uniform(-1500, 1500, 3)]
    engine0 = 1 + (seed * 2654435761) % 2147483645
    s0.hidden.valid |= 4; s0.hidden.ref_engine = engine0
    tape = np.zeros((ticks, nc, 8), np.float32)
    for k in range(nc):
        t = 0
        while t < ticks:
            span = int(rng.randint(4, 60))
            c = np.zeros(8, np.float32)
            c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
            c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
            tape[t:t + span, k] = c; t += span
    return team, s0, engine0, 1 + (seed * 7) % 60, tape


def aerial_case(ref, seed, ticks):
    """tools/random_tapes.py `aerial` (round 6): every car in the air around the ball, any orientation, spinning, clear of the others; (team, start, engine, rehash, tape)"""
    rng = np.random.RandomState(seed)
    team = 1 + seed % 3; nc = 2 * team
    k0 = ref.arena(team); ref.lib.ref_arena_reset_kickoff(k0, seed); s0 = ref.get_state(k0); ref.lib.ref_arena_free(k0)

    def rand_rot():
        q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    bp = np.array([rng.uniform(-2500, 2500), rng.uniform(-3500, 3500), rng.uniform(400, 1500)])
    s0.ball.pos[:] = [float(x) for x in bp]; s0.ball.vel[:] = [float(x) for x in rng.uniform(-900, 900, 3) * 1.0]
    for k in range(nc):
        c = s0.cars[k]
        for _ in range(200):
            R = rand_rot(); off = rng.normal(size=3); off *= rng.uniform(250, 900) / np.linalg.norm(off); pos = bp + off; pos[2] = min(max(pos[2], 150.0), 1750.0)
            if all(np.linalg.norm(pos - np.array(list(s0.cars[j].pos))) > 330.0 for j in range(k)): break
        c.pos[:] = [float(x) for x in pos]; c.rot[:] = [float(x) for x in np.concatenate([R[:, 0], R[:, 1], R[:, 2]])]
        c.vel[:] = [float(x) for x in (bp - pos) / np.linalg.norm(bp - pos) * rng.uniform(200, 1800) + rng.uniform(-200, 200, 3)]
        c.ang_vel[:] = [float(x) for x in rng.uniform(-4.5, 4.5, 3)]; c.flags = c.flags & ~0x1f
        c.boost = 100.0
    engine0 = 1 + (seed * 2654435761) % 2147483645
    tape = np.zeros((ticks, nc, 8), np.float32)
    for k in range(nc):
        t = 0
        while t < ticks:
            span = int(rng.randint(4, 60))
            c = np.zeros(8, np.float32)
            c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
            c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
            tape[t:t + span, k] = c; t += span
    return team, s0, engine0, 1 + (seed * 7) % 60, tape


def record(ref, team, start, engine0, rehash, tape, every=10):
    """the reference from `start` through the tape: its state every `every` ticks and the engine (layout of respawn_golden.npz's entries)"""
    L = ref.lib
    a = ref.arena(team)
    if rehash: L.ref_arena_rehash(a, rehash)
    ref.set_state(a, start)
    s0 = ArenaState.from_buffer_copy(bytes(start)); s0.car_order = ref.get_state(a).car_order
    L.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= 4; s0.hidden.ref_engine = engine0
    states, engines = [], []
    for t in range(len(tape)):
        for k in range(2 * team): ref.set_controls(a, k, tape[t][k])
        ref.step(a, 1)
        if (t + 1) % every == 0: states.append(state_vec(ref.get_state(a))); engines.append(L.ref_engine_state())
    L.ref_arena_free(a)
    return {"start_raw": np.frombuffer(bytes(s0), np.uint8).copy(), "tape": np.asarray(tape, np.float32), "states": np.asarray(states, np.float64), "engines": np.asarray(engines, np.uint32)}


def main():
    gold = np.load(os.path.join(HERE, "sim_golden.npz"))
    ref = RefSim(gold["mesh_verts"], gold["mesh_tris"]); L = ref.lib
    L.ref_arena_reset_kickoff.argtypes = [C.c_void_p, C.c_int]; L.ref_arena_free.argtypes = [C.c_void_p]; L.ref_arena_rehash.argtypes = [C.c_void_p, C.c_int]
    L.ref_engine_state.restype = C.c_uint32
    out = {"every": np.int32(10)}; names = []
    for seed in (9044,):
        team, s0, engine0, rehash, tape = wall_case(ref, seed, 400)
        rec = record(ref, team, s0, engine0, rehash, tape)
        name = f"walls_{seed}"; names.append(name)
        for k, v in rec.items(): out[f"phys/{name}/{k}"] = v
        print(name, f"{team}v{team}", len(tape), "ticks")
    # wheels "standing" on a car they do not touch: btSubsimplexConvexCast's hit when its 32 iterations run out, which the reference sees because its broadphase
    # hands a short ray every dynamic proxy on the ray cell's list (btRSBroadphase.cpp:326-337).  The start states carry RLGPU_MUT_RAY_PROXY_LISTS: with the switch
    # the stepper casts against the same bodies; without it (the product's default: the ray's box against the body's) these tapes leave at ticks 46 / 83 / 35.
    from rlgymppo_cpp_amd.state import HIDDEN_MUTATORS, MUT_RAY_PROXY_LISTS
    for seed, drawn in ((30118, 400), (80656, 300), (80142, 300), (80921, 300)):      # (the last: four cars on the ball in one tick -- see the file's head)      # (`drawn`: the tape length the tool was run with when it found the seed -- the draws depend on it)
        team, s0, engine0, rehash, tape = aerial_case(ref, seed, drawn)
        tape = tape[:300]
        s0.hidden.valid |= HIDDEN_MUTATORS; s0.mutators.flags |= MUT_RAY_PROXY_LISTS          # (s0 came from ref_arena_get_state: the block holds RLConst's defaults)
        rec = record(ref, team, s0, engine0, rehash, tape)
        name = f"aerial_{seed}"; names.append(name)
        for k, v in rec.items(): out[f"phys/{name}/{k}"] = v
        print(name, f"{team}v{team}", len(tape), "ticks")
    out["phys_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "edge_golden.npz"), **out)


if __name__ == "__main__":
    main()
