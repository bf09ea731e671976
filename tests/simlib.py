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
