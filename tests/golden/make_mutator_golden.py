#!/usr/bin/env python3
"""Fixtures that pin MutatorConfig's run-time scalars (round 6; VERDICT r05 "next" 6, the part that changes no shape, mass or material) against the REFERENCE.

Arena::SetMutatorConfig (RocketSim Arena.cpp:15-48) on the live reference's arena (oracle/ref_driver.cpp:ref_arena_set_mutators), then tapes that make every one of
those scalars matter, recorded every 10 ticks with the thread's random engine set to a known state (respawns draw from it: make_rng_golden.py):

  M1  gravity (120, -60, -325), car-world friction 0.15 / restitution 0.5, ball friction 0.2 / restitution 0.8, boost accel x 1.5 (ground) / x 0.6 (air), boost used x 0.5, jump accel x 1.25, immediate jump force x 0.8, ball max speed 2800, ball drag 0.12,
      respawn delay 1.5 s, bump cooldown 0.1 s, pad cooldowns 2.5 s / 1 s, spawn boost 61, ball-hit extra force x 1.6, bump force x 2, goal line 5000,
      unlimited flips AND double jumps, demolitions ON_CONTACT, team demolitions on
  M2  gravity -1000, car-world friction 0.6 / restitution 0.1, ball friction 0.9 / restitution 0.1 (the arena's own 0.6 / 0.3 win: btManifoldResult.cpp:56-78), boost accel x 0.5 / x 1.4, boost used x 2, jump accel x 0.7, immediate x 1.3, ball max speed 8000, ball drag 0, respawn delay 0.5 s, bump cooldown
      0.6 s, pad cooldowns 20 s / 8 s, spawn boost 5, ball-hit extra x 0.25, bump force x 0.4, demolitions DISABLED

  tapes per set: `charge` (2v2 head-on charges on full boost, then the hunt: bumps / demolitions / respawns / pads), `hunt3` (3v3 hunt from a kickoff), `spam`
  (1v1 and 2v2: a random action of the 90-row table every 8 ticks from a kickoff with full tanks: jumps, flips, double jumps, boost in the air, pads), `cannon`
  (1v1: the ball shot at 7000 uu/s across the field while the cars hunt it: max speed, drag, gravity, hits).

usage: python tests/golden/make_mutator_golden.py        (needs /root/reference built into oracle/_ref: make -C oracle ref)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from simlib import RefSim, state_vec  # noqa: E402
from rlgymppo_cpp_amd.state import (ArenaState, Mutators, HIDDEN_REF_ENGINE, HIDDEN_MUTATORS, MUT_UNLIMITED_FLIPS, MUT_UNLIMITED_DOUBLE_JUMPS,  # noqa: E402
                                    MUT_DEMO_ON_CONTACT, MUT_DEMO_DISABLED, MUT_TEAM_DEMOS)
from make_rng_golden import hunt_controls, charge_start, DEMOED  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
EVERY = 10


def mutator_sets():
    """(name, Mutators with everything but ball_damp_per_tick, ballDrag)"""
    d = dict(gravity_z=-650.0, boost_accel_ground=2975 / 3.0, boost_accel_air=3175 / 3.0, boost_used_per_second=100 / 3.0, jump_accel=4375 / 3.0, jump_immediate_force=875 / 3.0)
    m1 = Mutators(gravity_z=-325.0, boost_accel_ground=d["boost_accel_ground"] * 1.5, boost_accel_air=d["boost_accel_air"] * 0.6, boost_used_per_second=d["boost_used_per_second"] * 0.5,
                  jump_accel=d["jump_accel"] * 1.25, jump_immediate_force=d["jump_immediate_force"] * 0.8, ball_max_speed=2800.0, respawn_delay=1.5, bump_cooldown_time=0.1,
                  boost_pad_cooldown_big=2.5, boost_pad_cooldown_small=1.0, car_spawn_boost_amount=61.0, ball_hit_extra_force_scale=1.6, bump_force_scale=2.0, goal_base_threshold_y=5000.0,
                  gravity_x=120.0, gravity_y=-60.0, car_world_friction=0.15, car_world_restitution=0.5, ball_world_friction=0.2, ball_world_restitution=0.8,
                  flags=MUT_UNLIMITED_FLIPS | MUT_UNLIMITED_DOUBLE_JUMPS | MUT_DEMO_ON_CONTACT | MUT_TEAM_DEMOS)
    m2 = Mutators(gravity_z=-1000.0, boost_accel_ground=d["boost_accel_ground"] * 0.5, boost_accel_air=d["boost_accel_air"] * 1.4, boost_used_per_second=d["boost_used_per_second"] * 2.0,
                  jump_accel=d["jump_accel"] * 0.7, jump_immediate_force=d["jump_immediate_force"] * 1.3, ball_max_speed=8000.0, respawn_delay=0.5, bump_cooldown_time=0.6,
                  boost_pad_cooldown_big=20.0, boost_pad_cooldown_small=8.0, car_spawn_boost_amount=5.0, ball_hit_extra_force_scale=0.25, bump_force_scale=0.4, goal_base_threshold_y=5124.25,
                  gravity_x=0.0, gravity_y=0.0, car_world_friction=0.6, car_world_restitution=0.1, ball_world_friction=0.9, ball_world_restitution=0.1,
                  flags=MUT_DEMO_DISABLED)
    return [("M1", m1, 0.12), ("M2", m2, 0.0)]


def action_table():
    """DiscreteAction's 90 rows (throttle, steer, pitch, yaw, roll, jump, boost, handbrake): SIM/Utils/ActionParsers/DiscreteAction.cpp, as csrc/arena_gym.h builds it"""
    rows = []
    for throttle in (-1, 0, 1):
        for steer in (-1, 0, 1):
            for boost in (0, 1):
                for handbrake in (0, 1):
                    if boost == 1 and throttle != 1: continue
                    rows.append([throttle or boost, steer, 0, steer, 0, 0, boost, handbrake])
    for pitch in (-1, 0, 1):
        for yaw in (-1, 0, 1):
            for roll in (-1, 0, 1):
                for jump in (0, 1):
                    for boost in (0, 1):
                        if jump == 1 and yaw != 0: continue
                        if pitch == roll == jump == 0: continue
                        handbrake = jump == 1 and (pitch != 0 or yaw != 0 or roll != 0)
                        rows.append([boost, yaw, pitch, yaw, roll, jump, boost, int(handbrake)])
    return np.asarray(rows, np.float32)


def record(ref, team, start, mut, drag, engine0, rehash, n_ticks, controls):
    """`controls(t, current state) -> [nc][8]`; the arena runs under `mut`; returns the fixture entries"""
    nc = 2 * team
    a = ref.arena(team)
    if rehash: ref.lib.ref_arena_rehash(a, rehash)
    ref.lib.ref_arena_set_mutators(a, C.byref(mut), C.c_float(drag))
    ref.set_state(a, start)
    got = ref.get_state(a)
    s0 = ArenaState.from_buffer_copy(bytes(start)); s0.car_order = got.car_order
    s0.mutators = got.mutators; s0.hidden.valid |= HIDDEN_MUTATORS          # (the per-tick damping factor as the reference's C library forms it)
    ref.lib.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= HIDDEN_REF_ENGINE; s0.hidden.ref_engine = engine0
    tape, states, engines = [], [], []
    seen = dict(demos=0, respawns=0, max_ball_speed=0.0, pads_taken=0)
    was = [False] * nc; pads_was = None
    for t in range(n_ticks):
        cur = ref.get_state(a)
        ctl = controls(t, cur)
        for k in range(nc): ref.set_controls(a, k, ctl[k])
        ref.step(a, 1)
        tape.append(ctl)
        now = ref.get_state(a)
        dem = [bool(now.cars[k].flags & DEMOED) for k in range(nc)]
        seen["demos"] += sum(d and not w for d, w in zip(dem, was)); seen["respawns"] += sum(w and not d for d, w in zip(dem, was)); was = dem
        seen["max_ball_speed"] = max(seen["max_ball_speed"], float(np.linalg.norm(list(now.ball.vel))))
        pads = [bool(now.pads[p].is_active) for p in range(34)]
        if pads_was is not None: seen["pads_taken"] += sum(w and not p for p, w in zip(pads, pads_was))
        pads_was = pads
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
    L.ref_engine_state.restype = C.c_uint32; L.ref_arena_set_mutators.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
    table = action_table()
    assert table.shape == (90, 8)

    def kickoff(team, seed, boost=100.0):
        k0 = ref.arena(team); L.ref_arena_reset_kickoff(k0, seed); s = ref.get_state(k0); L.ref_arena_free(k0)
        for k in range(2 * team): s.cars[k].boost = boost
        return s

    out = {"every": np.int32(EVERY)}; names = []
    for mname, mut, drag in mutator_sets():
        cases = []
        # 2v2 head-on charges, then the hunt
        def charge_then_hunt(t, cur):
            c = hunt_controls(cur, 4)
            if t < 90: c[:, 1] = 0.0; c[:, 6] = 1.0; c[:, 7] = 0.0      # the charge: straight ahead on full boost
            return c
        cases.append(("charge", 2, charge_start(ref, 2, 301), 7, 1500, charge_then_hunt))
        cases.append(("hunt3", 3, kickoff(3, 302), 13, 1500, lambda t, cur: hunt_controls(cur, 6)))
        for team, seed in ((1, 303), (2, 304)):
            rng = np.random.RandomState(seed)
            acts = rng.randint(0, 90, size=(400, 2 * team))
            cases.append((f"spam{team}", team, kickoff(team, seed), 5 * team, 1600, lambda t, cur, acts=acts, nc=2 * team: table[acts[t // 8, :nc]].copy()))
        s = kickoff(1, 305); s.ball.pos[:] = [-3000.0, -2000.0, 400.0]; s.ball.vel[:] = [5200.0, 4300.0, 1900.0]
        cases.append(("cannon", 1, s, 0, 1200, lambda t, cur: hunt_ball(cur)))
        for cname, team, start, rehash, n_ticks, ctl in cases:
            name = f"{mname}/{cname}"
            rec, seen = record(ref, team, start, mut, drag, 1 + (sum(map(ord, name)) * 7919 + team * 104729 + n_ticks) % 2147483000, rehash, n_ticks, ctl)
            names.append(name)
            for k, v in rec.items(): out[f"phys/{name}/{k}"] = v
            print(f"{name}: {n_ticks} ticks: {seen}")
    out["phys_names"] = np.array(names)

    # ---- Gym rollouts under M1 (the layout of sim_golden.npz's gym/ entries: the tests' rollout bodies take either file) ----------------------------------
    # the gym layer reads two of the scalars itself: GameEventTracker asks Arena::IsBallScored (goal line 5000 + radius, not RLGymSim's own 5124.25) and
    # Arena::IsBallProbablyGoingIn (gravity, goal line) for goals / shots / saves, while GoalScoreCondition keeps RLGymSim's constant (Math.cpp:3-5)
    from simlib import RefGym
    from rlgymppo_cpp_amd.state import default_arena
    tab = np.zeros((128, 8), np.float32); n = L.ref_action_table(tab.ctypes.data_as(C.c_void_p), 128); tab = tab[:n]
    assert np.array_equal(tab, table), "the generator's action table is not the reference's"
    idle = int(np.argmin(np.abs(tab).sum(1))); assert np.abs(tab[idle]).sum() == 0
    mname, mut, drag = mutator_sets()[0]
    gcases = []
    s = default_arena(2); s.ball.pos[:] = (200, 4300, 93.15); s.ball.vel[:] = (0, 1500, 0)
    gcases.append(("M1_1v1_goal_line", 1, 8, 0, 2, 150, s, np.full((60, 2), idle, np.int32)))
    s = default_arena(4); s.ball.pos[:] = (-600, 2500, 600); s.ball.vel[:] = (300, 1900, 500)
    for k in range(4): s.cars[k].boost = 100
    gcases.append(("M1_2v2_random", 2, 8, 0, 3, 150, s, np.random.RandomState(11).randint(0, 90, size=(90, 4)).astype(np.int32)))
    gnames = []
    for case, team, tick_skip, omp, rk, nts, s0, acts in gcases:
        g = RefGym(ref, team, tick_skip, reward_kind=rk, no_touch_steps=nts, obs_max_players=omp)
        L.ref_arena_set_mutators(g.arena(), C.byref(mut), C.c_float(drag))
        obs0 = g.reset_to(s0)
        start = ref.get_state(g.arena())
        s0.car_order = start.car_order; s0.mutators = start.mutators; s0.hidden.valid |= HIDDEN_MUTATORS
        engine0 = 1 + (sum(map(ord, case)) * 7919 + team) % 2147483000       # the respawns of the rollout draw from the thread's engine (make_rng_golden.py)
        L.ref_seed_engine(C.c_uint32(engine0)); s0.hidden.valid |= HIDDEN_REF_ENGINE; s0.hidden.ref_engine = engine0
        obs = []; rew = []; done = []; order = []; last = None
        for t in range(len(acts)):
            o, r, d, st = g.step(acts[t])
            obs.append(o); rew.append(r); done.append(d); order.append(g.player_order()); last = st
            if d:
                acts = acts[: t + 1]
                break
        out[f"gym/{case}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()
        out[f"gym/{case}/obs0"] = obs0; out[f"gym/{case}/actions"] = acts
        out[f"gym/{case}/obs"] = np.stack(obs); out[f"gym/{case}/rew"] = np.stack(rew); out[f"gym/{case}/done"] = np.array(done, np.int32)
        out[f"gym/{case}/player_order"] = np.array(order, np.int32)
        out[f"gym/{case}/final"] = np.frombuffer(bytes(last), np.uint8).copy()
        out[f"gym/{case}/cfg"] = np.array([team, tick_skip, omp, rk, nts], np.int32)
        fin = last
        print(f"{case}: {len(acts)} steps, done {done[-1]}, reward sums {np.stack(rew).sum(0)}, score line {list(fin.gym.score_line)}, "
              f"goals/shots/saves {[(fin.gym.players[k].match_goals, fin.gym.players[k].match_shots, fin.gym.players[k].match_saves) for k in range(2 * team)]}")
        gnames.append(case)
    out["gym_names"] = np.array(gnames); out["mesh_verts"] = gold["mesh_verts"]; out["mesh_tris"] = gold["mesh_tris"]
    np.savez_compressed(os.path.join(HERE, "mutator_golden.npz"), **out)


def hunt_ball(cur):
    """both cars drive at the ball on full boost, jumping when it is above them"""
    out = np.zeros((2, 8), np.float32)
    for k in range(2):
        me = cur.cars[k]
        dx, dy = cur.ball.pos[0] - me.pos[0], cur.ball.pos[1] - me.pos[1]
        fx, fy = me.rot[0], me.rot[1]
        ang = float(np.arctan2(fx * dy - fy * dx, fx * dx + fy * dy))
        out[k] = [1.0, float(np.clip(-2.0 * ang, -1.0, 1.0)), 0, 0, 0, 1.0 if (cur.ball.pos[2] > 250.0 and dx * dx + dy * dy < 600.0 ** 2) else 0.0, 1.0, 0.0]
    return out


if __name__ == "__main__":
    main()
