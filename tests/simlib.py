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
PORT_SO = os.path.join(ROOT, "oracle", "_build", "liboracle_port.so")

_vp = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_vp)


class PortSim:
    def __init__(self):
        self.lib = C.CDLL(PORT_SO)
        assert self.lib.port_state_size() == C.sizeof(ArenaState)

    def procedural_mesh(self):
        v = np.zeros((20000, 3), np.float32)
        t = np.zeros((20000, 3), np.int32)
        nv, nt = C.c_int(), C.c_int()
        rc = self.lib.port_procedural_mesh(_ptr(v), 20000, _ptr(t), 20000, C.byref(nv), C.byref(nt))
        assert rc == 0
        return v[: nv.value].copy(), t[: nt.value].copy()

    def set_mesh(self, verts, tris):
        verts = np.ascontiguousarray(verts, np.float32)
        tris = np.ascontiguousarray(tris, np.int32)
        self.lib.port_set_mesh(_ptr(verts), len(verts), _ptr(tris), len(tris))

    def step(self, state: ArenaState, ticks=1, seed=0, env=0):
        self.lib.port_arena_step(C.byref(state), ticks, C.c_uint32(seed), C.c_uint32(env))


class RefSim:
    _inited = False

    def __init__(self, verts, tris):
        self.lib = C.CDLL(REF_SO)
        assert self.lib.ref_state_size() == C.sizeof(ArenaState)
        self.lib.ref_arena_new.restype = _vp
        self.lib.ref_gym_new.restype = _vp
        self.lib.ref_gym_arena.restype = _vp
        self.lib.ref_bench_collect.restype = C.c_double
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


def port_gym_reset(port, states, cfg, run_setter=True):
    n = len(states); nc = states[0].num_cars
    D = 51 + 38 * cfg.obs_max_players if cfg.obs_max_players > 0 else 51 + 19 * nc
    arr = (ArenaState * n)(*states)
    obs = np.zeros((n * nc, D), np.float32)
    port.lib.port_gym_reset(arr, n, C.byref(cfg), _ptr(obs), 1 if run_setter else 0)
    return list(arr), obs


def port_gym_step(port, states, cfg, actions):
    n = len(states); nc = states[0].num_cars
    D = 51 + 38 * cfg.obs_max_players if cfg.obs_max_players > 0 else 51 + 19 * nc
    arr = (ArenaState * n)(*states)
    actions = np.ascontiguousarray(actions, np.int32)
    obs = np.zeros((n * nc, D), np.float32); rew = np.zeros(n * nc, np.float32); done = np.zeros(n * nc, np.int32)
    port.lib.port_gym_step(arr, n, C.byref(cfg), _ptr(actions), _ptr(obs), _ptr(rew), _ptr(done))
    return list(arr), obs, rew, done


class RefGym:
    def __init__(self, ref, team_size=1, tick_skip=8, obs_kind=0, reward_kind=0, no_touch_steps=150):
        self.ref = ref; self.nc = 2 * team_size; self.D = 51 + 19 * self.nc
        self.h = _vp(ref.lib.ref_gym_new(team_size, tick_skip, obs_kind, reward_kind, no_touch_steps))

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
