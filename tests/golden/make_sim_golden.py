"""Generates tests/golden/sim_golden.npz from the REAL reference (oracle/_ref/libref_oracle.so = RocketSim + RLGymSim_CPP
compiled from /root/reference by oracle/Makefile).  Run in the build container: `python tests/golden/make_sim_golden.py`.

Contents (data only):
  action_table                      DiscreteAction's 90x8 table
  phys/<scenario>/...               initial ArenaState bytes, per-tick control tape, reference states every `every` ticks
  gym/<case>/...                    initial state, action tape, per-step obs / reward / done of the reference Gym
The arena mesh is the repo's procedural soccar stand-in (the game's collision_meshes are not redistributable and are
absent here); it is stored too so the fixtures are self-contained.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from simlib import PortSim, RefSim, RefGym  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState, default_arena, yaw_rot, euler_rot, CF_BALLHIT_VALID  # noqa: E402

Z = [0.0] * 8


def scenarios():
    s = default_arena(2)
    out = {}
    out["rest"] = (s, lambda t, k: Z, 120)
    out["throttle"] = (s, lambda t, k: [1, 0, 0, 0, 0, 0, 0, 0] if k == 0 else Z, 240)
    out["steer_powerslide"] = (s, lambda t, k: [1, 1, 0, 1, 0, 0, 0, 0] if k == 0 else [1, -1, 0, -1, 0, 0, 1, 1], 240)
    out["jump"] = (s, lambda t, k: [0, 0, 0, 0, 0, 1 if t < 20 else 0, 0, 0], 200)
    out["flip"] = (s, lambda t, k: [1, 0, -1 if 30 <= t < 40 else 0, 0, 0, 1 if (t < 5 or 30 <= t < 35) else 0, 0, 0], 300)
    out["double_jump"] = (s, lambda t, k: [0, 0, 0, 0, 0, 1 if (t < 5 or 30 <= t < 35) else 0, 0, 0], 300)
    out["boost_turn"] = (s, lambda t, k: [1, 0.3, 0, 0.3, 0, 0, 1, 0], 400)
    s2 = default_arena(2); s2.ball.pos[:] = (100, 200, 600); s2.ball.vel[:] = (300, -200, 10); s2.ball.ang_vel[:] = (1, 2, -1)
    out["ball_drop"] = (s2, lambda t, k: Z, 600)
    s2 = default_arena(2); s2.ball.pos[:] = (100, 200, 93.15); s2.ball.vel[:] = (800, -500, 0)
    out["ball_roll"] = (s2, lambda t, k: Z, 600)
    s2 = default_arena(2); s2.ball.pos[:] = (0, -1000, 93.15)
    out["car_hits_ball"] = (s2, lambda t, k: [1, 0, 0, 0, 0, 0, 1, 0] if k == 0 else Z, 300)
    s2 = default_arena(2); s2.ball.pos[:] = (3000, 0, 300); s2.ball.vel[:] = (2500, 300, 200)
    out["ball_side_wall"] = (s2, lambda t, k: Z, 300)
    s2 = default_arena(2); s2.ball.pos[:] = (2000, 4000, 300); s2.ball.vel[:] = (100, 2500, 200)
    out["ball_back_wall_mesh"] = (s2, lambda t, k: Z, 300)
    s2 = default_arena(2); s2.ball.pos[:] = (3000, 4000, 200); s2.ball.vel[:] = (1800, 1900, -300)
    out["ball_corner_fillets"] = (s2, lambda t, k: Z, 400)
    s2 = default_arena(2); s2.ball.pos[:] = (300, 4500, 200); s2.ball.vel[:] = (500, 2500, 300)
    out["ball_into_goal"] = (s2, lambda t, k: Z, 400)
    s2 = default_arena(2); s2.cars[0].pos[:] = (0, -2000, 800); s2.cars[0].flags = 0; s2.cars[0].vel[:] = (100, 300, 200)
    out["air_control"] = (s2, lambda t, k: [1, 0, 0.7, -0.5, 1 if t < 60 else 0, 0, 1 if t % 40 < 20 else 0, 0] if k == 0 else Z, 220)
    s2 = default_arena(2); s2.cars[0].pos[:] = (3200, 0, 17); s2.cars[0].rot[:] = yaw_rot(0.0); s2.cars[0].boost = 100
    out["wall_ramp"] = (s2, lambda t, k: [1, 0, 0, 0, 0, 0, 1, 0] if k == 0 else Z, 300)
    s2 = default_arena(2); s2.cars[0].pos[:] = (0, -1000, 17); s2.cars[1].pos[:] = (20, 1000, 17); s2.cars[0].boost = 100; s2.cars[1].boost = 100
    out["car_car_head_on"] = (s2, lambda t, k: [1, 0, 0, 0, 0, 0, 1, 0], 300)
    s2 = default_arena(2); s2.cars[0].pos[:] = (0, -2000, 100); s2.cars[0].flags = 0; s2.cars[0].rot[:] = euler_rot(1.0, 0, 3.1)
    out["roof_landing_autoflip"] = (s2, lambda t, k: [1, 0, 0, 0, 0, 1 if 200 <= t < 210 else 0, 0, 0] if k == 0 else Z, 400)
    s2 = default_arena(2); s2.cars[0].pos[:] = (-3584, -200, 17); s2.cars[0].boost = 10
    out["boost_pad_pickup"] = (s2, lambda t, k: [1, 0, 0, 0, 0, 0, 0, 0] if k == 0 else Z, 240)
    return out


def extra_scenarios():
    """Round 2: hitbox-vs-mesh, car-car bump / demo / respawn, ball pinches, 2v2 and 3v3."""
    out = {}
    B = [1, 0, 0, 0, 0, 0, 1, 0]   # throttle + boost

    def arena(nc, cars):
        s = default_arena(nc)
        for k, (pos, yaw, boost) in enumerate(cars):
            s.cars[k].pos[:] = pos; s.cars[k].rot[:] = yaw_rot(yaw); s.cars[k].boost = boost
        return s
    # hitbox into the back-wall mesh, head on and at an angle (box-triangle narrowphase)
    s = arena(2, [((2500, 4200, 17), np.pi / 2, 100), ((-2500, -4200, 17), -np.pi / 2 + 0.5, 100)])
    out["car_into_back_wall"] = (s, lambda t, k: B, 260)
    # into the 45-degree corner wall, and along it
    s = arena(2, [((3300, 3500, 17), np.pi / 4, 100), ((-3400, -3300, 17), -np.pi / 2 - 0.3, 100)])
    out["car_into_corner_wall"] = (s, lambda t, k: B, 300)
    # goal: post (convex edge), side wall of the goal box, back of the net
    s = arena(2, [((700, 4300, 17), np.pi / 2 + 0.12, 100), ((-300, -4400, 17), -np.pi / 2 - 0.25, 100)])
    out["car_into_goal"] = (s, lambda t, k: B, 300)
    # x = +4096 wall plane at an angle; floor fillet under the wheels first
    s = arena(2, [((3000, 500, 17), 0.35, 100), ((-3000, -500, 17), np.pi - 0.6, 60)])
    out["car_into_side_wall"] = (s, lambda t, k: B, 300)
    # jump + flip into the ceiling region is too slow; drop a car from the ceiling upside down instead, spinning
    s = arena(2, [((500, -500, 1900), 0.3, 50), ((-800, 900, 1200), -1.0, 50)])
    s.cars[0].flags = 0; s.cars[0].rot[:] = euler_rot(0.3, 0.4, 2.9); s.cars[0].ang_vel[:] = (1.0, -2.0, 0.5); s.cars[0].vel[:] = (200, 100, 600)
    s.cars[1].flags = 0; s.cars[1].rot[:] = euler_rot(-1.0, -1.2, 0.7); s.cars[1].ang_vel[:] = (-3.0, 1.0, 2.5); s.cars[1].vel[:] = (-300, 50, -900)
    out["tumbling_drops"] = (s, lambda t, k: [0, 0, 0.3 if k == 0 else -0.5, 0.2, 0.4 if t % 60 < 30 else -0.4, 0, 0, 0], 400)
    # supersonic demo + respawn (3 s) of the victim; the attacker drives on
    s = arena(2, [((0, -3000, 17), np.pi / 2, 100), ((0, 1500, 17), -np.pi / 2, 0)])
    out["demo_and_respawn"] = (s, lambda t, k: B if k == 0 else Z, 620)
    # side-on bump (not supersonic): victim standing across the attacker's path
    s = arena(2, [((0, -1200, 17), np.pi / 2, 30), ((10, 0, 17), 0.0, 0)])
    out["side_bump"] = (s, lambda t, k: [1, 0, 0, 0, 0, 0, 1 if t < 40 else 0, 0] if k == 0 else Z, 300)
    # ball pinched between a car and the back wall / floor
    s = arena(2, [((1500, 3200, 17), np.pi / 2, 100), ((0, -3000, 17), np.pi / 2, 0)])
    s.ball.pos[:] = (1500, 4500, 93.15)
    out["ball_pinch_back_wall"] = (s, lambda t, k: B if k == 0 else Z, 260)
    # dribble: ball dropped on the roof of a moving car
    s = arena(2, [((0, -2000, 17), np.pi / 2, 100), ((2000, 2000, 17), -np.pi / 2, 0)])
    s.cars[0].vel[:] = (0, 600, 0); s.ball.pos[:] = (0, -1980, 160); s.ball.vel[:] = (0, 600, 0)
    out["ball_on_roof"] = (s, lambda t, k: [0.6, 0.05, 0, 0.05, 0, 0, 0, 0] if k == 0 else Z, 300)
    # aerial hit: jump + boost into a falling ball
    s = arena(2, [((0, -1500, 17), np.pi / 2, 100), ((0, 2500, 17), -np.pi / 2, 100)])
    s.ball.pos[:] = (0, -300, 700); s.ball.vel[:] = (0, -100, 0)
    out["aerial_hit"] = (s, lambda t, k: [1, 0, -0.6 if 20 <= t < 60 else 0, 0, 0, 1 if t < 25 else 0, 1, 0] if k == 0 else Z, 260)
    # 2v2: two cars chase the ball from each side, one of them jumping
    s = arena(4, [((-300, -1500, 17), np.pi / 2, 100), ((300, 1500, 17), -np.pi / 2, 100), ((400, -2200, 17), np.pi / 2 + 0.2, 60), ((-500, 2300, 17), -np.pi / 2 + 0.15, 60)])
    out["2v2_ball_chase"] = (s, lambda t, k: [1, 0.1 if k == 0 else (-0.1 if k == 1 else 0), 0, 0, 0, 1 if (k == 2 and 40 <= t < 50) else 0, 1 if t < 120 else 0, 0], 400)
    # 3v3 kickoff: everybody boosts at the ball
    s = arena(6, [((-2048, -2560, 17), np.pi / 4, 33.3), ((2048, 2560, 17), -3 * np.pi / 4, 33.3), ((2048, -2560, 17), 3 * np.pi / 4, 33.3),
                  ((-2048, 2560, 17), -np.pi / 4, 33.3), ((0, -4608, 17), np.pi / 2, 33.3), ((0, 4608, 17), -np.pi / 2, 33.3)])
    out["3v3_kickoff"] = (s, lambda t, k: [1, 0, 0, 0, 0, 1 if (k == 4 and 100 <= t < 110) else 0, 1, 0], 360)
    return out


def state_vec(s: ArenaState):
    v = list(s.ball.pos) + list(s.ball.vel) + list(s.ball.ang_vel)
    for k in range(s.num_cars):
        c = s.cars[k]
        v += list(c.pos) + list(c.vel) + list(c.ang_vel) + list(c.rot) + [float(c.flags), c.boost]
    return np.array(v, np.float32)


def gym_cases(ref):
    """(name, team_size, tick_skip, obs_max_players, reward_kind, no_touch_steps, start state, actions [steps][nc])."""
    tab = np.zeros((128, 8), np.float32)
    n = ref.lib.ref_action_table(tab.ctypes.data_as(C.c_void_p), 128); tab = tab[:n]

    def act(thr=0, steer=0, pitch=0, yaw=0, roll=0, jump=0, boost=0, hb=0):
        d = np.abs(tab - np.array([thr, steer, pitch, yaw, roll, jump, boost, hb], np.float32)).sum(1)
        i = int(np.argmin(d)); assert d[i] == 0
        return i
    FWD, IDLE = act(thr=1, boost=1), act()

    def arena(nc, cars):
        s = default_arena(nc)
        for k, (pos, yaw, boost) in enumerate(cars):
            s.cars[k].pos[:] = pos; s.cars[k].rot[:] = yaw_rot(yaw); s.cars[k].boost = boost
        return s
    out = []
    s0 = default_arena(2); s0.ball.pos[:] = (0, -1500, 93.15); s0.cars[0].boost = 100
    for case, ts, seed in [("ts8_random", 8, 0), ("ts8_chase", 8, 1), ("ts1_random", 1, 2)]:
        rng = np.random.RandomState(seed)
        steps = 160 if ts == 8 else 60
        acts = rng.randint(0, 90, size=(steps, 2)).astype(np.int32)
        if case == "ts8_chase":
            acts[:40, 0] = 21
        out.append((case, 1, ts, 0, 0, 150, s0, acts))
    # NoTouchCondition fires (nobody reaches the ball in 12 steps); GoalScoreCondition never evaluated after it (Match.cpp:32-38)
    out.append(("1v1_timeout", 1, 8, 0, 0, 12, s0, np.random.RandomState(3).randint(0, 90, size=(40, 2)).astype(np.int32)))
    # a ball rolling into the orange goal: GoalScoreCondition + EventReward{teamGoal, concede}
    s = default_arena(2); s.ball.pos[:] = (200, 4300, 93.15); s.ball.vel[:] = (0, 1500, 0)
    out.append(("1v1_goal", 1, 8, 0, 0, 150, s, np.full((60, 2), IDLE, np.int32)))
    # goal + shot for blue 2, assist + shot pass for blue 0 (GameEventTracker.cpp:5-116), every reward term
    s = arena(4, [((-1500, 3900, 17), 0.0, 100), ((3000, -3000, 17), -np.pi / 2, 0), ((0, 2300, 17), np.pi / 2, 100), ((-3000, -3000, 17), -np.pi / 2, 0)])
    s.ball.pos[:] = (-1150, 3900, 93.15)
    a = np.full((60, 4), IDLE, np.int32); a[:6, 0] = FWD; a[2:, 2] = FWD
    out.append(("2v2_goal_assist_allterms", 2, 8, 0, 2, 150, s, a))
    # shot by orange 1 (its touch 40 ticks before the start), save by blue 0, who then bumps and demolishes orange 1; zero-sum of every term
    s = arena(4, [((0, -3600, 17), np.pi / 2, 100), ((0, 1500, 17), -np.pi / 2, 100), ((2500, 2000, 17), np.pi / 2, 0), ((-2500, 2000, 17), -np.pi / 2, 0)])
    s.tick_count = 1000; s.ball_update_counter = 1000
    s.ball.pos[:] = (0, -600, 93.15); s.ball.vel[:] = (0, -2600, 0)
    s.cars[1].flags |= CF_BALLHIT_VALID; s.cars[1].bh_tick_hit = 960; s.cars[1].bh_tick_extra = 960
    a = np.full((60, 4), IDLE, np.int32); a[:, 0] = FWD
    out.append(("2v2_shot_save_demo_zerosum", 2, 8, 0, 3, 150, s, a))
    # 2v2, DefaultOBSPadded(3) (one zero block per list, shuffled), zero-sum of every term, random actions around the ball
    s = arena(4, [((-400, -900, 17), np.pi / 2, 60), ((300, 800, 17), -np.pi / 2, 60), ((500, -1500, 17), np.pi / 2 + 0.3, 40), ((-600, 1400, 17), -np.pi / 2 - 0.2, 40)])
    rng = np.random.RandomState(4); a = rng.randint(0, 90, size=(100, 4)).astype(np.int32); a[:25] = FWD
    out.append(("2v2_padded3_zerosum_random", 2, 8, 3, 3, 150, s, a))
    # 3v3, DefaultOBS(165), every term, from the kickoff layout
    s = arena(6, [((-2048, -2560, 17), np.pi / 4, 33.3), ((2048, 2560, 17), -3 * np.pi / 4, 33.3), ((2048, -2560, 17), 3 * np.pi / 4, 33.3),
                  ((-2048, 2560, 17), -np.pi / 4, 33.3), ((0, -4608, 17), np.pi / 2, 33.3), ((0, 4608, 17), -np.pi / 2, 33.3)])
    rng = np.random.RandomState(5); a = rng.randint(0, 90, size=(90, 6)).astype(np.int32); a[:10] = FWD
    out.append(("3v3_allterms_random", 3, 8, 0, 2, 150, s, a))
    return out


def one_team_cases(ref):
    """Gyms WITHOUT opponents -- Match(..., spawnOpponents = false) -- as (name, team_size, tick_skip, obs_max_players, reward_kind,
    no_touch_steps, start state, actions [steps][team_size]).  The start state has 2 * team_size slots, the blue cars on the even ones."""
    from rlgymppo_cpp_amd.state import CF_ABSENT, CF_IS_DEMOED
    tab = np.zeros((128, 8), np.float32)
    n = ref.lib.ref_action_table(tab.ctypes.data_as(C.c_void_p), 128); tab = tab[:n]

    def act(thr=0, steer=0, boost=0):
        d = np.abs(tab - np.array([thr, steer, 0, steer, 0, 0, boost, 0], np.float32)).sum(1)
        i = int(np.argmin(d)); assert d[i] == 0
        return i
    FWD = act(thr=1, boost=1)

    def arena(team, cars):
        s = default_arena(2 * team)
        for k in range(1, 2 * team, 2):
            s.cars[k].flags = CF_IS_DEMOED | CF_ABSENT; s.cars[k].demo_respawn_timer = 1e30; s.cars[k].pos[:] = (0, 0, -10000)
        for j, (pos, yaw, boost) in enumerate(cars):
            s.cars[2 * j].pos[:] = pos; s.cars[2 * j].rot[:] = yaw_rot(yaw); s.cars[2 * j].boost = boost
        return s
    out = []
    # one car pushes the ball into the orange goal: touch, shot, goal; EventReward{teamGoal}, GoalScoreCondition
    s = arena(1, [((0, 3000, 17), np.pi / 2, 100)]); s.ball.pos[:] = (0, 3900, 93.15)
    out.append(("1v0_push_into_goal", 1, 8, 0, 0, 150, s, np.full((80, 1), FWD, np.int32)))
    # nobody touches the ball: NoTouchCondition ends the episode
    s = arena(1, [((2000, -3000, 17), 0.0, 30)]); s.ball.pos[:] = (-1000, 2000, 93.15)
    out.append(("1v0_timeout", 1, 8, 0, 0, 10, s, np.random.RandomState(11).randint(0, 90, size=(30, 1)).astype(np.int32)))
    # two team mates, every reward term inside ZeroSumReward (the opponents' mean is that of an empty team), random play around the ball
    s = arena(2, [((-300, -1000, 17), np.pi / 2, 80), ((400, -1500, 17), np.pi / 2 + 0.2, 50)]); s.ball.pos[:] = (0, 0, 93.15)
    a = np.random.RandomState(12).randint(0, 90, size=(70, 2)).astype(np.int32); a[:22] = FWD
    out.append(("2v0_allterms_zerosum_random", 2, 8, 0, 3, 150, s, a))
    # three team mates, DefaultOBSPadded(3): two teammate blocks and three zero opponent blocks, shuffled
    # (not the mirror-symmetric kickoff spots: two cars meeting exactly head-on at the centre line is a coin toss between which of them
    # bumps the other, decided by the last bit of the first contact point)
    s = arena(3, [((-2048, -2560, 17), np.pi / 4, 33.3), ((1500, -3300, 17), 2.0, 60.0), ((0, -4608, 17), np.pi / 2, 33.3)])
    a = np.random.RandomState(13).randint(0, 90, size=(60, 3)).astype(np.int32); a[:20] = FWD
    out.append(("3v0_padded3_allterms", 3, 8, 3, 2, 150, s, a))
    return out


def main_one_team():
    """tests/golden/sim_golden_one_team.npz: the one-team gym rollouts of the reference (kept apart from sim_golden.npz so that file stays as it is)."""
    port = PortSim()
    verts, tris = port.procedural_mesh()
    ref = RefSim(verts, tris)
    out = {"mesh_verts": verts, "mesh_tris": tris}
    gnames = []
    for case, team, tick_skip, omp, rk, nts, s0, acts in one_team_cases(ref):
        g = RefGym(ref, team, tick_skip, reward_kind=rk, no_touch_steps=nts, obs_max_players=omp, spawn_opponents=False)
        obs0 = g.reset_to(s0)
        start = ref.get_state(g.arena())
        obs = []; rew = []; done = []; order = []; last = None
        for t in range(len(acts)):
            o, r, d, st = g.step(acts[t])
            obs.append(o); rew.append(r); done.append(d); order.append(g.player_order()); last = st
            if d:
                acts = acts[: t + 1]
                break
        out[f"gym/{case}/start"] = np.frombuffer(bytes(start), np.uint8).copy()
        s0.car_order = start.car_order
        out[f"gym/{case}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()
        out[f"gym/{case}/obs0"] = obs0
        out[f"gym/{case}/actions"] = acts
        out[f"gym/{case}/obs"] = np.stack(obs); out[f"gym/{case}/rew"] = np.stack(rew); out[f"gym/{case}/done"] = np.array(done, np.int32)
        out[f"gym/{case}/player_order"] = np.array(order, np.int32)
        out[f"gym/{case}/final"] = np.frombuffer(bytes(last), np.uint8).copy()
        out[f"gym/{case}/cfg"] = np.array([team, tick_skip, omp, rk, nts, 0], np.int32)     # last entry: spawnOpponents
        gnames.append(case)
        print(case, "steps", len(acts), "done", done[-1], "reward sum", np.sum(rew))
    out["gym_names"] = np.array(gnames)
    np.savez_compressed(os.path.join(HERE, "sim_golden_one_team.npz"), **out)
    print("wrote sim_golden_one_team.npz:", len(gnames), "gym rollouts")


def main():
    port = PortSim()
    verts, tris = port.procedural_mesh()
    ref = RefSim(verts, tris)
    out = {"mesh_verts": verts, "mesh_tris": tris}
    tab = np.zeros((128, 8), np.float32)
    n = ref.lib.ref_action_table(tab.ctypes.data_as(C.c_void_p), 128)
    out["action_table"] = tab[:n].copy()
    every = 10
    names = []
    sc = scenarios(); sc.update(extra_scenarios())
    steps_before = {2: [], 4: [], 6: []}; steps_after = {2: [], 4: [], 6: []}; steps_tag = {2: [], 4: [], 6: []}
    mbuf = np.zeros((64, 16), np.float32)
    for si, (name, (s0, fn, ticks)) in enumerate(sc.items()):
        nc = s0.num_cars
        a = ref.arena(nc // 2)
        ref.set_state(a, s0)
        start = ref.get_state(a)
        s0.car_order = start.car_order     # the arena's per-car loop order (an accident of heap addresses in the reference) travels with the state
        pair_arena = ref.arena(nc // 2)   # one-tick pairs are stepped from the RECORDED state (set_state) in an arena of their own
        tape = np.zeros((ticks, nc, 8), np.float32)
        rec = []
        for t in range(ticks):
            for k in range(nc):
                tape[t, k] = fn(t, k)
                ref.set_controls(a, k, list(tape[t, k]))
            before = ref.get_state(a)
            ref.step(a, 1)
            after = ref.get_state(a)
            # one-step pairs: every tick on which the reference's narrowphase produced a contact, and every 16th tick
            if ref.lib.ref_debug_manifolds(a, mbuf.ctypes.data_as(C.c_void_p), 64) > 0 or t % 16 == 0:
                # The free-running arena above never leaves Bullet units; a state read from it is rounded to uu and back once more than
                # its own next tick sees.  So the pair's "after" is what the reference computes FROM the recorded "before": both sides of
                # the comparison then start from the same bits (grounded cars and a flying ball come out bit-equal that way).
                # (One pair arena per tape, stepped through the tape's pairs in turn: what a state does not carry -- the wheels' previous-tick
                # raycast records -- is then what the previous contact tick left, as in the free-running arena.  Its car order is its own:
                # the pair's "before" is given the pair arena's order.)
                ref.set_state(pair_arena, before)
                before.car_order = ref.get_state(pair_arena).car_order
                for k in range(nc):
                    ref.set_controls(pair_arena, k, list(tape[t, k]))
                ref.step(pair_arena, 1)
                after_pair = ref.get_state(pair_arena)
                steps_before[nc].append(np.frombuffer(bytes(before), np.uint8).copy()); steps_after[nc].append(np.frombuffer(bytes(after_pair), np.uint8).copy())
                steps_tag[nc].append((si, t))
            if (t + 1) % every == 0:
                rec.append(state_vec(after))
        out[f"phys/{name}/start"] = np.frombuffer(bytes(start), np.uint8).copy()
        out[f"phys/{name}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()   # what set_state was given: `start` is that state read back (once through Bullet units)
        out[f"phys/{name}/tape"] = tape
        out[f"phys/{name}/states"] = np.stack(rec)
        names.append(name)
        ref.lib.ref_arena_free(a); ref.lib.ref_arena_free(pair_arena)
    out["phys_names"] = np.array(names)
    out["phys_every"] = every

    gnames = []
    for case, team, tick_skip, omp, rk, nts, s0, acts in gym_cases(ref):
        g = RefGym(ref, team, tick_skip, reward_kind=rk, no_touch_steps=nts, obs_max_players=omp)
        obs0 = g.reset_to(s0)
        start = ref.get_state(g.arena())
        obs = []; rew = []; done = []; order = []; last = None
        for t in range(len(acts)):
            o, r, d, st = g.step(acts[t])
            obs.append(o); rew.append(r); done.append(d); order.append(g.player_order()); last = st
            if d:
                acts = acts[: t + 1]
                break
        out[f"gym/{case}/start"] = np.frombuffer(bytes(start), np.uint8).copy()
        s0.car_order = start.car_order   # what the gym was reset to (the rollout continues from THAT, not from the copy read back): see add_gym_start_raw.py
        out[f"gym/{case}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()
        out[f"gym/{case}/obs0"] = obs0
        out[f"gym/{case}/actions"] = acts
        out[f"gym/{case}/obs"] = np.stack(obs); out[f"gym/{case}/rew"] = np.stack(rew); out[f"gym/{case}/done"] = np.array(done, np.int32)
        out[f"gym/{case}/player_order"] = np.array(order, np.int32)
        out[f"gym/{case}/final"] = np.frombuffer(bytes(last), np.uint8).copy()     # arena + gym-level carried state after the last step
        out[f"gym/{case}/cfg"] = np.array([team, tick_skip, omp, rk, nts], np.int32)
        gnames.append(case)
    out["gym_names"] = np.array(gnames)

    # state setters (A8): the reference's own RandomState(true, true, true) and KickoffState, 4000 resets each, as data
    for team in (1, 2, 3):
        for kind, kname in ((0, "random"), (1, "kickoff")):
            nset = 4000 if kind == 0 else 600
            arr = (ArenaState * nset)()
            ref.lib.ref_setter_samples(team, kind, nset, arr)
            rows = np.stack([np.concatenate([state_vec(x)[:9]] + [np.concatenate([state_vec(x)[9 + 20 * k: 9 + 20 * k + 18], [x.cars[k].boost]]) for k in range(2 * team)]) for x in arr])
            out[f"setter/{kname}/team{team}"] = rows.astype(np.float32)      # ball 9, per car pos3 vel3 ang3 rot9 boost
    np.savez_compressed(os.path.join(HERE, "sim_golden.npz"), **out)
    st = {}
    for nc in (2, 4, 6):
        if steps_before[nc]:
            st[f"nc{nc}/before"] = np.stack(steps_before[nc]); st[f"nc{nc}/after"] = np.stack(steps_after[nc]); st[f"nc{nc}/tag"] = np.array(steps_tag[nc], np.int32)
    st["phys_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "sim_steps.npz"), **st)
    print("wrote sim_golden.npz:", len(names), "physics scenarios,", len(gnames), "gym rollouts; sim_steps.npz:", {k: len(v) for k, v in steps_tag.items()}, "one-tick pairs")


if __name__ == "__main__" and "--one-team" in sys.argv:
    main_one_team()
elif __name__ == "__main__":
    main()
