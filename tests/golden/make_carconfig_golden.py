#!/usr/bin/env python3
"""Fixtures that pin CarConfig (round 6; VERDICT r05 "next" 6, the hitbox / wheel half) against the REFERENCE.

Arenas of the live reference whose cars are built from the five presets other than the Octane (Arena::AddCar(team, CAR_CONFIG_*), RocketSim CarConfig.cpp:7-79;
oracle/ref_driver.cpp:ref_arena_new_cfg), then tapes in which the hitbox, its offset, the box's inertia, the wheel positions, radii and rest lengths all matter,
recorded every 10 ticks with the thread's random engine set to a known state (respawns draw from it: make_rng_golden.py):

  spam1 / spam2  1v1 / 2v2: a random row of the 90-row action table every 8 ticks from a kickoff with full tanks (driving, jumps, flips, landings on wheels and on
                 the hitbox, ball hits)
  charge         2v2 head-on charges on full boost, then the hunt (car-car box-box contacts, bumps, demolitions, respawns)
  hunt3          3v3 hunt from a kickoff
  cannon         1v1: the ball shot across the field while both cars chase it, jumping under it (car-ball GJK on every face of the hitbox)
  walls          1v1: both cars driven into the side wall and up it on full boost, then released (wheels and hitbox on the wall's triangles, falls onto the roof)

also the presets' own numbers as the reference holds them (`configs`, 17 floats per preset: the repo's table in csrc/arena_io.h is checked against them).

usage: python tests/golden/make_carconfig_golden.py        (needs /root/reference built into oracle/_ref: make -C oracle ref)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from simlib import RefSim, state_vec  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState, HIDDEN_REF_ENGINE  # noqa: E402
from make_rng_golden import hunt_controls, charge_start, DEMOED  # noqa: E402
from make_mutator_golden import action_table, hunt_ball  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
EVERY = 10
PRESETS = ["OCTANE", "DOMINUS", "PLANK", "BREAKOUT", "HYBRID", "MERC"]


def record(ref, preset, team, start, engine0, rehash, n_ticks, controls):
    nc = 2 * team
    a = C.c_void_p(ref.lib.ref_arena_new_cfg(team, preset))
    if rehash: ref.lib.ref_arena_rehash(a, rehash)
    # one tick before the tape's start state goes in: a Bullet world that has never stepped holds btContactSolverInfo's default m_timeStep = 1 / 60
    # (btContactSolverInfo.h:82; set to the tick's 1 / 120 by the first stepSimulation, btDiscreteDynamicsWorld.cpp:417), and the wheels' pushback
    # (resolveSingleCollision, btVehicleRL.cpp:118-212) divides by it BEFORE that first stepSimulation: tick 1 of a never-stepped arena is not a tick of a game
    ref.set_state(a, start); ref.step(a, 1)
    ref.set_state(a, start)
    got = ref.get_state(a)
    s0 = ArenaState.from_buffer_copy(bytes(start)); s0.car_order = got.car_order
    ref.lib.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= HIDDEN_REF_ENGINE; s0.hidden.ref_engine = engine0
    tape, states, engines = [], [], []
    seen = dict(demos=0, respawns=0, ball_touches=0)
    was = [False] * nc; last_ball_vel = None
    for t in range(n_ticks):
        cur = ref.get_state(a)
        ctl = controls(t, cur)
        for k in range(nc): ref.set_controls(a, k, ctl[k])
        ref.step(a, 1)
        tape.append(ctl)
        now = ref.get_state(a)
        dem = [bool(now.cars[k].flags & DEMOED) for k in range(nc)]
        seen["demos"] += sum(d and not w for d, w in zip(dem, was)); seen["respawns"] += sum(w and not d for d, w in zip(dem, was)); was = dem
        bv = np.asarray(list(now.ball.vel))
        if last_ball_vel is not None and np.linalg.norm(bv - last_ball_vel) > 60.0: seen["ball_touches"] += 1
        last_ball_vel = bv
        if (t + 1) % EVERY == 0:
            states.append(np.concatenate([state_vec(now), [float(now.pads[p].is_active) for p in range(34)], [now.pads[p].cooldown for p in range(34)]]))
            engines.append(ref.lib.ref_engine_state())
    ref.lib.ref_arena_free(a)
    return {"start_raw": np.frombuffer(bytes(s0), np.uint8).copy(), "tape": np.asarray(tape, np.float32), "states": np.asarray(states, np.float64),
            "engines": np.asarray(engines, np.uint32)}, seen


def main():
    gold = np.load(os.path.join(HERE, "sim_golden.npz"))
    ref = RefSim(gold["mesh_verts"], gold["mesh_tris"])
    L = ref.lib
    L.ref_arena_reset_kickoff.argtypes = [C.c_void_p, C.c_int]; L.ref_arena_free.argtypes = [C.c_void_p]; L.ref_arena_rehash.argtypes = [C.c_void_p, C.c_int]
    L.ref_engine_state.restype = C.c_uint32; L.ref_arena_new_cfg.restype = C.c_void_p; L.ref_arena_new_cfg.argtypes = [C.c_int, C.c_int]
    L.ref_car_config.argtypes = [C.c_int, C.c_void_p]
    table = action_table()

    def kickoff(team, seed, boost=100.0):
        k0 = ref.arena(team); L.ref_arena_reset_kickoff(k0, seed); s = ref.get_state(k0); L.ref_arena_free(k0)
        for k in range(2 * team): s.cars[k].boost = boost
        return s

    configs = np.zeros((6, 17), np.float32)
    for p in range(6): L.ref_car_config(p, configs[p].ctypes.data)
    out = {"every": np.int32(EVERY), "configs": configs, "preset_names": np.asarray(PRESETS)}; names = []
    for preset in range(1, 6):
        cases = []
        for team, seed in ((1, 403), (2, 404)):
            rng = np.random.RandomState(seed + 10 * preset)
            acts = rng.randint(0, 90, size=(400, 2 * team))
            cases.append((f"spam{team}", team, kickoff(team, seed + preset), 5 * team, 1200, lambda t, cur, acts=acts, nc=2 * team: table[acts[t // 8, :nc]].copy()))

        def charge_then_hunt(t, cur):
            c = hunt_controls(cur, 4)
            if t < 90: c[:, 1] = 0.0; c[:, 6] = 1.0; c[:, 7] = 0.0
            return c
        cases.append(("charge", 2, charge_start(ref, 2, 411 + preset), 7, 1300, charge_then_hunt))
        cases.append(("hunt3", 3, kickoff(3, 420 + preset), 13, 1200, lambda t, cur: hunt_controls(cur, 6)))
        s = kickoff(1, 430 + preset); s.ball.pos[:] = [-3000.0, -2000.0, 400.0]; s.ball.vel[:] = [3200.0, 2300.0, 1500.0]
        cases.append(("cannon", 1, s, 0, 1000, lambda t, cur: hunt_ball(cur)))
        # walls: blue towards +x, orange towards -x, 1500 uu/s on the ground, facing the side walls
        s = kickoff(1, 440 + preset)
        for k, sx in ((0, 1.0), (1, -1.0)):
            c = s.cars[k]
            c.pos[:] = [sx * 2200.0, -1500.0 + 3000.0 * k, 17.0]; c.vel[:] = [sx * 1500.0, 0.0, 0.0]; c.ang_vel[:] = [0.0, 0.0, 0.0]
            c.rot[:] = [sx, 0.0, 0.0, 0.0, sx, 0.0, 0.0, 0.0, 1.0]     # forward, right, up
        s.ball.pos[:] = [0.0, 0.0, 93.15]; s.ball.vel[:] = [0.0, 0.0, 0.0]

        def walls(t, cur):
            c = np.zeros((2, 8), np.float32)
            if t < 260: c[:, 0] = 1.0; c[:, 6] = 1.0
            elif t < 420: c[:, 0] = 1.0; c[:, 1] = 0.6
            else: c[:, 0] = -1.0; c[:, 4] = 1.0; c[:, 2] = 0.3
            return c
        cases.append(("walls", 1, s, 0, 800, walls))
        for cname, team, start, rehash, n_ticks, ctl in cases:
            name = f"{PRESETS[preset]}/{cname}"
            rec, seen = record(ref, preset, team, start, 1 + (sum(map(ord, name)) * 7919 + team * 104729 + n_ticks) % 2147483000, rehash, n_ticks, ctl)
            print(name, seen, flush=True)
            for k, v in rec.items(): out[f"phys/{name}/{k}"] = v
            out[f"phys/{name}/preset"] = np.int32(preset)
            names.append(name)
    out["phys_names"] = np.asarray(names)
    np.savez_compressed(os.path.join(HERE, "carconfig_golden.npz"), **out)
    print("wrote carconfig_golden.npz:", len(names), "tapes")


if __name__ == "__main__":
    main()
