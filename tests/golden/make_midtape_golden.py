"""Generates tests/golden/midtape_golden.npz from the REAL reference: states taken MID-EPISODE together with the arena's hidden state.

An arena holds more than its CarStates and BallState: btRSBroadphase remembers where each dynamic proxy was last filed and in which order the
proxies arrived in their cells -- that order is the order of the overlapping pairs, of the manifolds, of the solver's rows -- and a demolished
car's rigid body has a basis of its own.  RlgpuArenaState::hidden carries both across the C-ABI (include/rlgpu_state.h); the reference's side is
read by oracle/ref_driver.cpp:ref_arena_get_hidden from the broadphase's cell lists.

For the tapes of sim_golden.npz in which several bodies meet (and two of the wedge fixture's on its 16-file mesh is left out: one mesh per process):
the reference arena runs the tape to tick T1; S = its state there; the SAME arena is set to S (so that it continues from the values S holds --
a state is in uu, the arena steps in Bullet units: only a state that was set can be continued bit for bit) and keeps its broadphase history;
S.hidden = that history; then the tape runs on to its end, recorded every 10 ticks.  The stepper, given S in a FRESH env, must continue the same
way; given S without the hidden block it is a fresh arena set to S, which is another thing (the tests assert that at least one tape tells the two
apart).

    python tests/golden/make_midtape_golden.py          (build container)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

from simlib import RefSim, state_vec  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState  # noqa: E402

EVERY = 10
CUTS = {"3v3_kickoff": [200, 280, 300, 330], "demo_and_respawn": [60, 150], "car_car_head_on": [40], "side_bump": [60], "2v2_ball_chase": [150], "ball_pinch_back_wall": [50]}


def main():
    sg = np.load(os.path.join(HERE, "sim_golden.npz"))
    ref = RefSim(sg["mesh_verts"], sg["mesh_tris"])
    ref.lib.ref_arena_get_hidden.argtypes = [C.c_void_p, C.c_void_p]; ref.lib.ref_arena_free.argtypes = [C.c_void_p]
    out = {"mesh_verts": sg["mesh_verts"], "mesh_tris": sg["mesh_tris"]}
    names = []
    for name, cuts in CUTS.items():
        tape = sg[f"phys/{name}/tape"]
        for t1 in cuts:
            if t1 >= len(tape) - 2 * EVERY: continue
            s0 = ArenaState.from_buffer_copy(sg[f"phys/{name}/start_raw"].tobytes()); nc = s0.num_cars
            a = ref.arena(nc // 2); ref.set_state(a, s0)
            if ref.get_state(a).car_order != s0.car_order:    # (the arena's per-car loop order is an accident of heap addresses: try again)
                for _ in range(64):
                    ref.lib.ref_arena_free(a); a = ref.arena(nc // 2); ref.set_state(a, s0)
                    if ref.get_state(a).car_order == s0.car_order: break
            order = ref.get_state(a).car_order
            for t in range(t1):
                for k in range(nc): ref.set_controls(a, k, list(tape[t, k]))
                ref.step(a, 1)
            s = ref.get_state(a)
            ref.set_state(a, s)                               # the same arena, continued from the values the state holds
            ref.lib.ref_arena_get_hidden(a, C.byref(s))
            s.car_order = order
            rec = []
            for t in range(t1, len(tape)):
                for k in range(nc): ref.set_controls(a, k, list(tape[t, k]))
                ref.step(a, 1)
                if (t + 1 - t1) % EVERY == 0: rec.append(state_vec(ref.get_state(a)))
            key = f"{name}@{t1}"
            out[f"cut/{key}/state"] = np.frombuffer(bytes(s), np.uint8).copy()
            out[f"cut/{key}/tape"] = tape[t1:].astype(np.float32)
            out[f"cut/{key}/states"] = np.stack(rec)
            names.append(key)
            hist = [int(x) for x in s.hidden.bp_hist[:nc + 1]]
            print(f"{key}: hist cells {[h >> 3 for h in hist]} ranks {[h & 7 for h in hist]}, demolished {[k for k in range(nc) if s.cars[k].flags & 0x400]}, {len(rec)} records", flush=True)
            ref.lib.ref_arena_free(a)
    out["names"] = np.array(names); out["every"] = np.int32(EVERY)
    path = os.path.join(HERE, "midtape_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes:", len(names), "cuts")


if __name__ == "__main__":
    main()
