"""Shared ctypes loaders for the test-side simulators (TEST INFRASTRUCTURE).

  RefSim  -> oracle/_ref/libref_oracle.so : the real reference (RocketSim + RLGymSim_CPP), when built
  PortSim -> oracle/_build/liboracle_port.so : host build of the stepper core
Both speak rlgymppo_cpp_amd.state.ArenaState in slot order.
"""
import ctypes as C
import os
import numpy as np

from rlgymppo_cpp_amd.state import ArenaState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so")
PORT_SO = os.environ.get("RLG_PORT_SO") or os.path.join(ROOT, "oracle", "_build", "liboracle_port.so")     # (override: development builds of the port for tools/raw_divergence.py)

_vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_vp)


class PortSim:
    def __init__(self, so=None):
        self.lib = C.CDLL(so or PORT_SO)
        assert self.lib.port_state_size() == C.sizeof(ArenaState)

    def procedural_mesh(self):
        v = np.zeros((20000, 3), np.float32)
        t = np.zeros((20000, 3), np.int32)
        nv, nt = C.c_int(), C.c_int()
        rc = self.lib.port_procedural_mesh(_ptr(v), 20000, _ptr(t), 20000, C.byref(nv), C.byref(nt))
        assert rc == 0
        return v[: nv.value].copy(), t[: nt.value].copy()

    def set_mesh(self, verts, tris, parts=None):
        """parts: triangles per mesh object (.cmf file) in input order, or None = one object."""
        verts = np.ascontiguousarray(verts, np.float32)
        tris = np.ascontiguousarray(tris, np.int32)
        if parts is None:
            self.lib.port_set_mesh(_ptr(verts), len(verts), _ptr(tris), len(tris))
        else:
            parts = np.ascontiguousarray(parts, np.int32)
            self.lib.port_set_mesh_parts(_ptr(verts), len(verts), _ptr(tris), len(tris), _ptr(parts), len(parts))

    def step(self, state: ArenaState, ticks=1, seed=0, env=0, hist=None):
        """hist: a (C.c_uint16 * 8)() the caller keeps per arena -- the broadphase's memory of its proxies, which no state carries; without it
        every call steps a FRESH arena set to `state` (like BatchedEnv, whose env slots keep theirs across uploads)."""
        if hist is not None:
            self.lib.port_step_hist(C.byref(state), ticks, C.byref(hist))
        else:
            self.lib.port_arena_step(C.byref(state), ticks, C.c_uint32(seed), C.c_uint32(env))


class RefSim:
    _inited = False

    def __init__(self, verts, tris, mesh_dir=None):
        """mesh_dir: a directory holding soccar/*.cmf, loaded by the reference's own RocketSim::Init (one collision object per file);
        else the one-blob mesh (verts, tris).  A process initialises the reference once."""
        self.lib = C.CDLL(REF_SO)
        assert self.lib.ref_state_size() == C.sizeof(ArenaState)
        self.lib.ref_arena_new.restype = _vp
        self.lib.ref_gym_new.restype = _vp
        self.lib.ref_gym_arena.restype = _vp
        self.lib.ref_bench_collect.restype = C.c_double
        if mesh_dir is not None:
            rc = self.lib.ref_init_dir(str(mesh_dir).encode())
        else:
            verts = np.ascontiguousarray(verts, np.float32)
            tris = np.ascontiguousarray(tris, np.int32)
            rc = self.lib.ref_init(_ptr(verts), len(verts), _ptr(tris), len(tris))
        assert rc == 0

    def arena(self, team_size=1):
        return _vp(self.lib.ref_arena_new(team_size))

    def set_state(self, a, s):
        self.lib.ref_arena_set_state(a, C.byref(s))

    def get_state(self, a):
        s = ArenaState()
        self.lib.ref_arena_get_state(a, C.byref(s))
        return s

    def set_controls(self, a, slot, c8):
        arr = (C.c_float * 8)(*c8)
        self.lib.ref_arena_set_controls(a, slot, arr)

    def step(self, a, ticks=1):
        self.lib.ref_arena_step(a, ticks)


def have_ref():
    return os.path.exists(REF_SO)


def have_port():
    return os.path.exists(PORT_SO)


# ---- gym-level helpers -----------------------------------------------------------------------------------------
def port_gym_cfg(**over):
    """A GymConfig (same layout as the C-ABI's) with the examplemain.cpp defaults; pure ctypes, no GPU library."""
    from rlgymppo_cpp_amd._lib import GymConfig, RW_FACE_BALL, RW_VEL_PLAYER_TO_BALL, RW_VEL_BALL_TO_GOAL, RW_EVENT
    c = GymConfig()
    c.tick_skip = 8
    c.n_terms = 4
    for i, (k, w) in enumerate([(RW_FACE_BALL, 0.1), (RW_VEL_PLAYER_TO_BALL, 0.5), (RW_VEL_BALL_TO_GOAL, 1.0), (RW_EVENT, 50.0)]):
        c.terms[i].kind = k; c.terms[i].weight = w; c.terms[i].p0 = 0.0
    c.event_weights[1] = 1.0; c.event_weights[2] = -1.0
    c.zero_sum = 0; c.team_spirit = 0.0; c.opp_scale = 1.0
    c.n_conds = 2; c.conds[0] = 0; c.conds[1] = 1; c.no_touch_max_steps = 150
    c.setter_kind = 0; c.rand_ball_speed = 1; c.rand_car_speed = 1; c.cars_on_ground = 1
    c.seed_lo = 123; c.seed_hi = 0
    c.pos_coef[0] = 1 / 4096.0; c.pos_coef[1] = 1 / 5120.0; c.pos_coef[2] = 1 / 2044.0
    c.vel_coef = 1 / 2300.0; c.ang_vel_coef = 1 / 5.5
    c.n_actions = 90
    for k, v in over.items():
        setattr(c, k, v)
    return c


def all_terms_cfg(zero_sum=False, **over):
    """GymConfig mirroring ref_gym_new2's reward_kind 2 / 3 (oracle/ref_driver.cpp): every CommonRewards.h term, distinct weights."""
    from rlgymppo_cpp_amd._lib import RW_EVENT, RW_VELOCITY, RW_SAVE_BOOST, RW_VEL_BALL_TO_GOAL, RW_VEL_PLAYER_TO_BALL, RW_FACE_BALL, RW_TOUCH_BALL
    c = port_gym_cfg(**over)
    terms = [(RW_EVENT, 10.0, 0.0), (RW_VELOCITY, 0.11, 0.0), (RW_SAVE_BOOST, 0.07, 0.5), (RW_VEL_BALL_TO_GOAL, 0.9, 0.0),
             (RW_VEL_PLAYER_TO_BALL, 0.45, 0.0), (RW_FACE_BALL, 0.13, 0.0), (RW_TOUCH_BALL, 0.8, 0.7)]
    c.n_terms = len(terms)
    for i, (k, w, p0) in enumerate(terms):
        c.terms[i].kind = k; c.terms[i].weight = w; c.terms[i].p0 = p0
    for i, w in enumerate([1.0, 0.5, -0.75, 0.6, 0.05, 0.3, 0.2, 0.4, 0.35, -0.25, 0.15]):
        c.event_weights[i] = w
    if zero_sum:
        c.zero_sum = 1; c.team_spirit = 0.3; c.opp_scale = 0.8
    return c


def _gym_rows(cfg, nc):
    rows = nc // 2 if cfg.one_team else nc
    return rows, (51 + 38 * cfg.obs_max_players if cfg.obs_max_players > 0 else 51 + 19 * rows)


def port_gym_reset(port, states, cfg, run_setter=True):
    n = len(states); nc, D = _gym_rows(cfg, states[0].num_cars)
    arr = (ArenaState * n)(*states)
    obs = np.zeros((n * nc, D), np.float32)
    port.lib.port_gym_reset(arr, n, C.byref(cfg), _ptr(obs), 1 if run_setter else 0)
    return list(arr), obs


def port_gym_step(port, states, cfg, actions, hist=None):
    """hist: np.zeros((n, 8), np.uint16) kept by the caller across steps -- each arena's broadphase history (PortSim.step)."""
    n = len(states); nc, D = _gym_rows(cfg, states[0].num_cars)
    arr = (ArenaState * n)(*states)
    actions = np.ascontiguousarray(actions, np.int32)
    obs = np.zeros((n * nc, D), np.float32); rew = np.zeros(n * nc, np.float32); done = np.zeros(n * nc, np.int32)
    port.lib.port_gym_step_hist(arr, n, C.byref(cfg), _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done), _ptr(hist) if hist is not None else None)
    return list(arr), obs, rew, done


class RefGym:
    """The reference's Gym behind oracle/ref_driver.cpp.  obs_max_players 0: DefaultOBS, m: DefaultOBSPadded(m); reward_kind 0/1: example
    stack plain / zero-sum, 2/3: every CommonRewards term plain / zero-sum (ref_gym_new2)."""
    def __init__(self, ref, team_size=1, tick_skip=8, obs_kind=0, reward_kind=0, no_touch_steps=150, obs_max_players=0, spawn_opponents=True):
        self.ref = ref; self.nc = 2 * team_size if spawn_opponents else team_size   # agent rows
        self.D = 51 + 38 * obs_max_players if obs_max_players > 0 else 51 + 19 * self.nc
        ref.lib.ref_gym_new3.restype = _vp
        self.h = _vp(ref.lib.ref_gym_new3(team_size, tick_skip, obs_max_players, reward_kind, no_touch_steps, 1 if spawn_opponents else 0))

    def player_order(self):
        out = (C.c_int32 * 8)()
        n = self.ref.lib.ref_gym_player_order(self.h, out)
        return [int(out[i]) - 1 for i in range(n)]     # agent rows (= slots with two teams)

    def reset_to(self, state):
        obs = np.zeros((self.nc, self.D), np.float32)
        d = self.ref.lib.ref_gym_reset_to(self.h, C.byref(state), _ptr(obs))
        assert d == self.D
        return obs

    def step(self, actions):
        actions = np.ascontiguousarray(actions, np.int32)
        obs = np.zeros((self.nc, self.D), np.float32); rew = np.zeros(self.nc, np.float32); done = C.c_int32()
        st = ArenaState()
        self.ref.lib.ref_gym_step(self.h, _ptr(actions), _ptr(obs), _ptr(rew), C.byref(done), C.byref(st))
        return obs, rew, int(done.value), st

    def arena(self):
        return _vp(self.ref.lib.ref_gym_arena(self.h))


# ---- comparison helpers shared by the CPU (port) and GPU (HIP) tests against the reference fixtures ------------------------------
def state_vec(s):
    """ball pos / vel / angvel (9), then per car pos3 vel3 angvel3 rot9 flags boost (20)."""
    v = list(s.ball.pos) + list(s.ball.vel) + list(s.ball.ang_vel)
    for k in range(s.num_cars):
        c = s.cars[k]
        v += list(c.pos) + list(c.vel) + list(c.ang_vel) + list(c.rot) + [float(c.flags), c.boost]
    return np.array(v, np.float64)


def phys_errors(got, ref, nc):
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    pos = max(d[0:3].max(), max(d[9 + 20 * k: 12 + 20 * k].max() for k in range(nc)))
    vel = max(d[3:6].max(), max(d[12 + 20 * k: 15 + 20 * k].max() for k in range(nc)))
    ang = max(d[6:9].max(), max(d[15 + 20 * k: 18 + 20 * k].max() for k in range(nc)))
    rot = max(d[18 + 20 * k: 27 + 20 * k].max() for k in range(nc))
    flags = any(got[27 + 20 * k] != ref[27 + 20 * k] for k in range(nc))
    return pos, vel, ang, rot, flags


def _tol(pos, vel, ang, rot, until=None):
    return {"pos": pos, "vel": vel, "ang": ang, "rot": rot, "until": until}


# Free-run tolerances (uu, uu/s, rad/s, rotation-matrix entries) over the WHOLE tape unless `until` is given: the loose table, kept for
# runs that start from the state READ BACK from the reference (phys/<name>/start: once through Bullet units) -- the exact statement is
# PHYS_EXACT_UNTIL below, for runs from the state set_state was given.  The restatement follows the reference's x86 arithmetic down to its
# rsqrtss-based normalize, the SSE summation orders of its quaternion / matrix code and of its row solver (rl_math.h, arena_step.h); the
# narrowphase routines -- GJK, the penetration-depth solver (second GJK + EPA, arena_epa.h), the internal-edge adjustment -- are
# bit-identical to Bullet's on fuzzed inputs.
PHYS_FREE_RUN = {
    "rest": _tol(0.005, 0.005, 1e-4, 1e-5), "throttle": _tol(0.01, 0.02, 1e-4, 1e-5), "steer_powerslide": _tol(0.03, 0.02, 1e-3, 1e-4),
    "jump": _tol(0.01, 0.05, 1e-3, 1e-4), "flip": _tol(0.02, 0.02, 1e-3, 1e-4), "double_jump": _tol(0.02, 0.05, 1e-3, 1e-4),
    "boost_turn": _tol(0.1, 0.05, 1e-3, 1e-4), "ball_drop": _tol(0.06, 0.02, 1e-3, 1e-4), "ball_roll": _tol(0.08, 0.02, 1e-3, 1e-4),
    "car_hits_ball": _tol(0.1, 0.1, 2e-3, 1e-4), "ball_side_wall": _tol(0.03, 0.05, 1e-3, 1e-4), "ball_back_wall_mesh": _tol(0.05, 0.05, 1e-3, 1e-4),
    "ball_corner_fillets": _tol(0.12, 0.03, 1e-3, 1e-4), "ball_into_goal": _tol(0.08, 0.05, 1e-3, 1e-4), "air_control": _tol(0.02, 0.02, 1e-3, 1e-4),
    "wall_ramp": _tol(0.15, 0.5, 0.01, 1e-3), "car_car_head_on": _tol(0.3, 0.4, 0.005, 5e-4), "roof_landing_autoflip": _tol(0.03, 0.02, 1e-3, 1e-4),
    "boost_pad_pickup": _tol(0.03, 0.02, 1e-3, 1e-4), "car_into_back_wall": _tol(0.3, 1.0, 0.02, 1e-3), "car_into_corner_wall": _tol(0.4, 1.0, 0.04, 0.002),
    "car_into_goal": _tol(0.1, 0.1, 1e-3, 1e-4, until=160), "car_into_side_wall": _tol(0.1, 0.1, 2e-3, 1e-4, until=290),
    "tumbling_drops": _tol(0.1, 0.3, 0.01, 0.003), "demo_and_respawn": _tol(1.0, 1.5, 0.1, 0.005, until=600), "side_bump": _tol(0.1, 0.1, 2e-3, 2e-4),
    "ball_pinch_back_wall": _tol(2.0, 4.0, 0.05, 0.01), "ball_on_roof": _tol(0.05, 0.05, 1e-3, 1e-4), "aerial_hit": _tol(0.03, 0.05, 1e-3, 1e-4),
    "2v2_ball_chase": _tol(0.15, 0.1, 2e-3, 1e-4), "3v3_kickoff": _tol(0.1, 0.1, 1e-3, 1e-4, until=230),
}
# Ticks for which a free run (inside the stepper's units, from the state the reference's set_state was given: phys/<name>/start_raw) is
# BIT-IDENTICAL to the reference's recorded trajectory, every field of every body; tapes not listed: their whole length.
PHYS_EXACT_UNTIL = {}     # empty since round 3: all 31 tapes over their whole length.  The last one, `3v3_kickoff` (a six-car heap with two
# demolitions, 360 ticks), needed the two pieces of state the reference keeps outside its CarStates: the order in which the dynamic proxies
# arrived in their broadphase cells (btRSBroadphase.cpp:185-203, 287-325, 393-469 -> arena_step.h bp_history_track) and the basis of a
# demolished car's rigid body, which stays in the world and swallows wheel rays (btDefaultVehicleRaycaster.cpp:36-51 -> arena_world.h
# car_ghost_rot).  Both live in the resident state and start as a fresh arena's when a state comes from the host.
# ... and how close a tape stays after that, until the given tick (pos uu, vel uu/s, ang rad/s, rot)
PHYS_AFTER_EXACT = {}

# One tick from the reference's own state: every one of the 1722 recorded pairs is bit-equal (asserted in the tests).  A scenario's pairs are
# stepped one after the other in ONE arena / env slot, as they were recorded: the broadphase's arrival order, which a state does not
# carry, passes from pair to pair (round 3; before that 14 pairs of the six-car heap were not exact and had a tolerance here).
ONE_TICK_TOL = {
    "default": {"pos": 0.0, "vel": 0.0},
}
ONE_TICK_NOT_EXACT_MAX = 0


# observation tolerance per gym fixture (default 2e-3 = 8 uu on a position, 4.6 uu/s on a velocity, 0.011 rad/s on an angular velocity)
GYM_OBS_TOL = {
    "2v2_shot_save_demo_zerosum": 1e-2,     # from the demolition on (step 45) the wreck's frozen angular velocity is 0.037 rad/s off
}


# gym rollouts of the reference that the HIP gym reproduces EXACTLY (resident state, no uu round trip between steps): every observation
# row and every reward bit-equal, dones and counters equal (tools/gym_fixture_errors.py).  Since round 3 that is ALL of them, the nine
# two-team rollouts and the four one-team ones.  Two things it took beyond the physics: the rollouts start from `gym/<case>/start_raw`,
# the state the reference's gym was reset to (`start` is that state read back, one rounding away, and the reference itself never continued
# from it); and SaveBoostReward / TouchBallReward call powf as the host libm rounds it (arena_gym.h libm_powf)
GYM_EXACT = {"ts8_random", "ts8_chase", "ts1_random", "1v1_timeout", "1v1_goal", "2v2_goal_assist_allterms", "2v2_shot_save_demo_zerosum",
             "2v2_padded3_zerosum_random", "3v3_allterms_random",
             "1v0_push_into_goal", "1v0_timeout", "2v0_allterms_zerosum_random", "3v0_padded3_allterms",
             "M1_1v1_goal_line", "M1_2v2_random"}     # (the last two: Gym rollouts under a non-default MutatorConfig, tests/golden/mutator_golden.npz)
GYM_EXACT_OBS = set()     # (rollouts with bit-equal observations but a last-bit reward difference: none left)

# steps up to which a free-running gym rollout is compared on the HIP path (resident arenas).  Empty since the manifold point's local point
# on the car comes from the detector's world point and not from the edge-adjusted one (round 3: `2v2_padded3_zerosum_random` had left at
# step 65; as tapes against the live reference both random rollouts are bit-identical in Bullet units to their end, tools/raw_divergence.py gym)
GYM_HORIZON = {}
# the host build's gym test hands the state over in uu after every step (one rounding per step the reference's resident arena does not
# make): its random 2v2 rollout with hitbox contacts is compared up to here
GYM_HORIZON_PORT = {"2v2_padded3_zerosum_random": 64, "M1_2v2_random": 64}


def gym_cfg_for_case(team, tick_skip, obs_max_players, reward_kind, no_touch_steps):
    """The GymConfig equivalent of oracle/ref_driver.cpp:ref_gym_new2(team, tick_skip, obs_max_players, reward_kind, no_touch_steps)."""
    if reward_kind >= 2:
        c = all_terms_cfg(zero_sum=(reward_kind == 3), tick_skip=tick_skip, no_touch_max_steps=no_touch_steps, obs_max_players=obs_max_players)
    else:
        c = port_gym_cfg(tick_skip=tick_skip, no_touch_max_steps=no_touch_steps, obs_max_players=obs_max_players)
        if reward_kind == 1:
            c.zero_sum = 1; c.team_spirit = 0.5; c.opp_scale = 1.0
    return c


def gym_compare_obs(got, ref, nc, obs_max_players, ref_order, tol, what, one_team=False):
    """Observation rows in slot order.  DefaultOBS lists the OTHER players in the reference's GameState::players order (an unordered_set
    iteration, recorded per step in the fixture) where this build uses slot order; DefaultOBSPadded shuffles both lists with a
    process-wide RNG: compared as the fixed part + the multiset of 19-float blocks per list."""
    got = np.asarray(got); ref = np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: obs shape {got.shape} vs {ref.shape}"
    assert np.abs(got[:, :70] - ref[:, :70]).max() < tol, f"{what}: ball / prev action / pads / self part differs by {np.abs(got[:, :70] - ref[:, :70]).max()}"
    rows = nc // 2 if one_team else nc                       # one-team gyms (spawnOpponents = false): agent rows = the blue cars, all teammates
    team = (lambda s: 0) if one_team else (lambda s: s % 2)
    for row in range(rows):
        mates = [s for s in ref_order if s != row and team(s) == team(row)]; opps = [s for s in ref_order if team(s) != team(row)]
        mine_m = [s for s in range(rows) if s != row and team(s) == team(row)]; mine_o = [s for s in range(rows) if team(s) != team(row)]
        if obs_max_players > 0:
            nm, no = obs_max_players - 1, obs_max_players
            for lo, n in ((70, nm), (70 + 19 * nm, no)):
                a = sorted(map(tuple, np.round(got[row, lo: lo + 19 * n].reshape(n, 19) / max(tol, 1e-6)).astype(np.int64) // 4))
                # sort-by-rounded-key is fragile near rounding boundaries: match blocks greedily instead
                G = list(got[row, lo: lo + 19 * n].reshape(n, 19)); R = list(ref[row, lo: lo + 19 * n].reshape(n, 19))
                for g in G:
                    j = int(np.argmin([np.abs(g - r).max() for r in R]))
                    assert np.abs(g - R[j]).max() < tol, f"{what}: row {row}: a padded block has no counterpart in the reference ({np.abs(g - R[j]).max()})"
                    R.pop(j)
        else:
            # this build: teammates in slot order, then opponents in slot order; the reference: the same lists in its players order
            for k, s in enumerate(mates + opps):
                mine = (mine_m + mine_o).index(s)
                a = got[row, 70 + 19 * mine: 70 + 19 * mine + 19]; b = ref[row, 70 + 19 * k: 70 + 19 * k + 19]
                assert np.abs(a - b).max() < tol, f"{what}: row {row}: block of player {s} differs by {np.abs(a - b).max()}"


def setter_samples_compare(got_states, ref, kind, nc, what):
    """One state setter's output (a list of ArenaState after a reset) against the reference's own samples (sim_golden.npz setter/*):
    kind 1 = KickoffState: per car exactly the reference's set of (x, y, z, yaw columns, boost), ball fixed; kind 0 = RandomState: every
    column inside the reference's support, quantiles within 5 % of the span (the reference's generator is seeded from the wall clock, so
    the comparison is statistical).  Shared by the CPU test (host setters) and the GPU test (the kernel's setters)."""
    got = np.stack([np.concatenate([state_vec(x)[:9]] + [np.concatenate([state_vec(x)[9 + 20 * k: 9 + 20 * k + 18], [x.cars[k].boost]]) for k in range(nc)]) for x in got_states])
    assert got.shape[1] == ref.shape[1], f"{what}: {got.shape} vs {ref.shape}"
    if kind == 1:
        for k in range(nc):
            cols = [9 + 19 * k + i for i in (0, 1, 2, 9, 10, 18)]
            a = {tuple((np.round(np.asarray(r, np.float64), 2) + 0.0).tolist()) for r in got[:, cols]}; b = {tuple((np.round(np.asarray(r, np.float64), 2) + 0.0).tolist()) for r in ref[:, cols]}
            assert a == b, f"{what} car {k}: spawn set differs: {sorted(a ^ b)[:4]}"
        assert np.abs(got[:, :9] - ref[0, :9]).max() < 1e-4
    else:
        qs = [0.5, 5, 25, 50, 75, 95, 99.5]
        for c in range(ref.shape[1]):
            lo, hi = float(ref[:, c].min()), float(ref[:, c].max()); span = max(hi - lo, 1e-3)
            # sample extremes of 4000 draws are noisy where the density thins out towards the end of the support: 10 % slack there
            assert got[:, c].min() >= lo - 0.1 * span - 1e-4 and got[:, c].max() <= hi + 0.1 * span + 1e-4, f"{what} col {c}: outside the reference's support"
            dq = np.abs(np.percentile(got[:, c], qs) - np.percentile(ref[:, c], qs)).max()
            assert dq < 0.05 * span + 1e-4, f"{what} col {c}: quantiles differ by {dq} (span {span})"


def write_cmf_parts(verts_uu, tris, parts, root):
    """A mesh of several objects as .cmf files under <root>/soccar/ (CollisionMeshFile.cpp:11-36: i32 nTris, i32 nVerts, index triplets,
    vertices in Bullet units), part_%02d.cmf in part order -- what RocketSim::Init / rlgpu_env_load_cmf_dir / ref_init_dir read."""
    d = os.path.join(root, "soccar"); os.makedirs(d, exist_ok=True)
    t0 = 0
    for k, n in enumerate(parts):
        n = int(n); tk = tris[t0:t0 + n]; t0 += n
        used = np.unique(tk); remap = -np.ones(len(verts_uu), np.int64); remap[used] = np.arange(len(used))
        with open(os.path.join(d, "part_%02d.cmf" % k), "wb") as f:
            f.write(np.int32(len(tk)).tobytes()); f.write(np.int32(len(used)).tobytes())
            f.write(remap[tk].astype(np.int32).tobytes()); f.write((verts_uu[used] * np.float32(0.02)).astype(np.float32).tobytes())
    return root


# the two-file fixture (tests/golden/seam_golden.npz): ticks for which a tape is bit-identical to the reference; not listed = its whole length.
# (Empty since the car's world-contact normal is the one the reference's callback sees -- the narrowphase normal BEFORE the internal-edge
# adjustment, Arena.cpp:218-282; `car_slides_along_panel` had left at tick 251, found with tools/raw_divergence.py.)
SEAM_EXACT_UNTIL = {}


def with_pads_of(state, current):
    """what a user state setter that sets ball and cars leaves behind: `state` with the boost pads of `current`"""
    from rlgymppo_cpp_amd.state import ArenaState
    out = ArenaState.from_buffer_copy(bytes(state))
    C.memmove(C.addressof(out.pads), C.addressof(current.pads), C.sizeof(out.pads))
    return out


# ---- GameInst::Step across episode ends (tests/golden/gameinst_golden.npz, recorded from the reference's own GameInst.cpp) -------------------
def gameinst_replay(gg, case, reset_to, step, obs_tol, rew_exact, what):
    """Replays one case of the fixture.  reset_to(state, first) -> obs rows [players][D] after a Gym::Reset whose state setter installs that
    state (the recorded setter's k-th call: ball, cars and tick count; the boost pads stay as the previous episode left them, which the new
    episode's first observation shows -- Match::ResetState resets them only after the setter returned, Match.cpp:55-69); step(actions) -> (obs rows, rewards, done) of one gym step, where the rows returned with done = 1 are not looked at: like
    GameInst::Step (GameInst.cpp:27-35) the replay then resets to the setter's next state and takes THOSE rows.  Compared with the reference:
    done and rewards every step, the rows of curObs every step (the first observation of the new episode where one ended), and the reward
    trackers -- curEpRew, avgEpRew, avgStepRew, totalSteps -- recomputed from the replay's own rewards the way GameInst.cpp:14-34 does."""
    from rlgymppo_cpp_amd.state import ArenaState
    team, tick_skip, omp, rk, nts = [int(x) for x in gg[f"gi/{case}/cfg"]]
    nc = 2 * team
    states = [ArenaState.from_buffer_copy(b.tobytes()) for b in gg[f"gi/{case}/states"]]
    acts = gg[f"gi/{case}/actions"]; cur = gg[f"gi/{case}/cur_obs"]; rew = gg[f"gi/{case}/rew"]; done = gg[f"gi/{case}/done"]
    trk = gg[f"gi/{case}/trackers"]; order = gg[f"gi/{case}/order"]; resets = gg[f"gi/{case}/resets"]
    k = 0
    obs = reset_to(states[k], True); k += 1
    gym_compare_obs(obs, cur[0], nc, omp, [int(x) for x in order[0]], obs_tol, f"{what} {case} start")
    f32 = np.float32
    cur_ep, ep_total, ep_count, st_total, st_count = f32(0), f32(0), 0, f32(0), 0
    ends = 0
    for t in range(len(acts)):
        o, r, d = step(acts[t])
        assert int(d) == int(done[t]), f"{what} {case}: done differs at step {t}"
        r = np.asarray(r, np.float32)
        if rew_exact:
            assert np.array_equal(r, rew[t]), f"{what} {case}: reward not bit-equal to the reference at step {t}: {r} vs {rew[t]}"
        else:
            assert np.abs(r - rew[t]).max() < 2e-3 * max(1.0, np.abs(rew[t]).max()), f"{what} {case}: reward differs at step {t}: {r} vs {rew[t]}"
        total = f32(0)
        for slot in order[t]:            # GameInst.cpp:16-18: in the players' order
            total = f32(total + r[int(slot)])
        st_total = f32(st_total + total); st_count += nc
        cur_ep = f32(cur_ep + f32(total / f32(nc)))
        if d:
            o = reset_to(states[k], False); k += 1; ends += 1
            ep_total = f32(ep_total + cur_ep); ep_count += 1; cur_ep = f32(0)
        assert k == int(resets[t + 1])
        gym_compare_obs(o, cur[t + 1], nc, omp, [int(x) for x in order[t + 1]], obs_tol, f"{what} {case} step {t}" + (" (first observation of the new episode)" if d else ""))
        got = np.array([cur_ep, ep_total, ep_count, st_total, st_count, t + 1], np.float32)
        if rew_exact:
            assert np.array_equal(got, trk[t]), f"{what} {case}: reward trackers at step {t}: {got} vs {trk[t]}"
        else:
            assert np.abs(got - trk[t]).max() < 2e-3 * max(1.0, np.abs(trk[t]).max()), f"{what} {case}: reward trackers at step {t}: {got} vs {trk[t]}"
    return ends


class OneCaseFixture(dict):
    """one gym rollout of an .npz in the layout of sim_golden.npz's gym/ entries, as the replaying tests read a fixture"""
    @property
    def files(self): return list(self.keys())


def live_gym_cases(rec, gold):
    """tests/golden/live_gym_golden.npz (tools/live_gym_hip.py --record): (case, one-case fixture, horizon) per rollout.  Every rollout is compared EXACTLY (GYM_EXACT).
    horizon: with DefaultOBSPadded the reference's obs builder shuffles with the engine Car::Respawn draws from (DefaultOBSPadded.cpp:58-59), which the stepper's
    replay of that engine does not follow (its shuffle is keyed Philox): such a rollout is comparable up to its first respawn."""
    for case in [str(x) for x in rec["gym_names"]]:
        entries = {k: rec[k] for k in rec.files if k.startswith(f"gym/{case}/")}
        omp = int(entries[f"gym/{case}/cfg"][2]); o = entries[f"gym/{case}/obs"]; horizon = len(o)
        if omp > 0:
            resp = [t for t in range(1, len(o)) if ((o[t - 1][:, 69] == 1) & (o[t][:, 69] == 0)).any()]
            if resp: horizon = resp[0]
        GYM_EXACT.add(case)
        if horizon < len(o): GYM_HORIZON[case] = GYM_HORIZON_PORT[case] = horizon
        yield case, OneCaseFixture({"gym_names": np.array([case]), "mesh_verts": gold["mesh_verts"], "mesh_tris": gold["mesh_tris"], **entries}), horizon
