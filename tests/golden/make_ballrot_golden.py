"""Generates tests/golden/ballrot_golden.npz from the REAL reference: tapes in which the ball starts with a basis that is NOT the identity
(BallState::rotMat set by a user state setter, Ball.cpp:41).  Under ArenaConfig::noBallRot (the default, ArenaConfig.h:33) the reference never
integrates the ball's orientation (btRigidBody.cpp:102-106), so the basis must come back unchanged from every GetState -- and it is the
frame a wheel's suspension ray is cast against the ball in (btCollisionWorld::rayTestSingle), which the first two tapes exercise: a car
dropped onto the ball, a car driving its front wheels up the ball.  The others let the ball roll into a car and along a wall.

    python tests/golden/make_ballrot_golden.py          (build container; a process of its own: the reference initialises once)

Contents (data only): mesh_verts / mesh_tris (the procedural arena), per scenario the start state (RlgpuArenaState bytes), the control tape,
the reference's state after EVERY tick (simlib.state_vec) and the ball basis it reported after every tick.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

from simlib import PortSim, RefSim, state_vec  # noqa: E402
from rlgymppo_cpp_amd.state import default_arena, yaw_rot, euler_rot  # noqa: E402

Z = [0.0] * 8


def scenarios():
    out = {}
    # a car dropped onto the resting ball: the wheels' rays meet the ball before anything else
    s = default_arena(2); s.hidden.ball_rot[:] = euler_rot(0.7, 0.3, -0.2)
    c = s.cars[0]; c.pos[:] = (10, -15, 93.15 + 92.75 + 45); c.rot[:] = yaw_rot(0.3); c.flags = 0
    out["car_dropped_on_rotated_ball"] = (s, lambda t, k: Z, 240)
    # a car driving its front wheels up the ball (throttle, no boost)
    s = default_arena(2); s.hidden.ball_rot[:] = euler_rot(-1.1, 0.9, 2.0)
    c = s.cars[0]; c.pos[:] = (0, -420, 17.0); c.rot[:] = yaw_rot(np.pi / 2)
    out["car_drives_up_rotated_ball"] = (s, lambda t, k: [0.35, 0, 0, 0, 0, 0, 0, 0] if k == 0 else Z, 360)
    # the ball (spinning, rotated basis) rolls into a standing car
    s = default_arena(2); s.hidden.ball_rot[:] = euler_rot(2.2, -0.4, 0.9)
    s.ball.pos[:] = (0, -1200, 93.15); s.ball.vel[:] = (30, -1100, 0); s.ball.ang_vel[:] = (3, -2, 1.5)
    out["rotated_ball_into_car"] = (s, lambda t, k: Z, 300)
    # ... and along the side wall's fillet and up the wall
    s = default_arena(2); s.hidden.ball_rot[:] = euler_rot(0.1, 1.3, -2.5)
    s.ball.pos[:] = (3200, 500, 300); s.ball.vel[:] = (1900, 700, -200); s.ball.ang_vel[:] = (-4, 2, 5)
    out["rotated_ball_side_wall"] = (s, lambda t, k: Z, 300)
    return out


def main():
    port = PortSim(); verts, tris = port.procedural_mesh(); port.set_mesh(verts, tris)
    ref = RefSim(verts, tris)
    out = {"mesh_verts": verts, "mesh_tris": tris, "names": np.array(list(scenarios().keys()))}
    for name, (s0, ctl, T) in scenarios().items():
        a = ref.arena(1); ref.set_state(a, s0)
        s0.car_order = ref.get_state(a).car_order
        tape = np.zeros((T, 2, 8), np.float32); states = []; rots = []
        wheel_on_ball = 0
        for t in range(T):
            for k in range(2):
                tape[t, k] = ctl(t, k); ref.set_controls(a, k, list(tape[t, k]))
            ref.step(a, 1)
            cur = ref.get_state(a)
            states.append(state_vec(cur)); rots.append(np.array(list(cur.hidden.ball_rot), np.float32))
        ref.lib.ref_arena_free(a)
        rots = np.stack(rots)
        assert np.array_equal(rots, np.tile(np.array(list(s0.hidden.ball_rot), np.float32), (T, 1))), f"{name}: the reference's ball basis changed"
        out[f"{name}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy(); out[f"{name}/tape"] = tape
        out[f"{name}/states"] = np.stack(states); out[f"{name}/ball_rot"] = rots
        # the host build on the same tape (resident in its own units between ticks, like the reference's arena), as a first check
        import ctypes as C
        st = type(s0).from_buffer_copy(bytes(s0)); outs = (type(s0) * T)()
        port.lib.port_run_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        tp = np.ascontiguousarray(tape, np.float32)
        port.lib.port_run_tape(C.byref(st), tp.ctypes.data, T, 1, C.byref(outs))
        bad = next((t + 1 for t in range(T) if not np.array_equal(state_vec(outs[t]), states[t])), None)
        rot_ok = all(list(outs[t].hidden.ball_rot) == list(s0.hidden.ball_rot) for t in range(T))
        print(f"{name}: {T} ticks, ball moved {np.abs(out[f'{name}/states'][-1][:3] - out[f'{name}/states'][0][:3]).max():.1f} uu, host build "
              + ("equal throughout" if bad is None else f"differs from tick {bad}") + (", basis kept" if rot_ok else ", BASIS CHANGED"))
    np.savez_compressed(os.path.join(HERE, "ballrot_golden.npz"), **out)


if __name__ == "__main__":
    main()
