"""Generates tests/golden/padreset_golden.npz from the REAL reference: what the first observation of a NEW episode shows of the boost pads when the
state setter is the reference's own RandomState (the example program's setter).  RandomState::ResetState starts with arena->ResetToRandomKickoff()
(StateSetters/RandomState.cpp:11), which resets every pad (Arena.cpp:209-210) before the setter builds the episode's first GameState -- so, unlike
with a user setter that only moves ball and cars (gameinst_golden.npz), the pads a car emptied in the previous episode are back in that
observation.  The setter's draws are not replayable (wall-clock-seeded std RNG), the pad part of the rows is: GameInst::Start / Step of the
reference's own GameInst.cpp, first episode from a chosen state (a car parked on a big pad and one on a small pad with room in their tanks), later
episodes from RandomState.  Run in the build container: `python tests/golden/make_padreset_golden.py`.

Per case <c>: pr/<c>/cfg = team, tick_skip, no_touch_steps;  start = ArenaState bytes;  actions [T][players] by slot;  pads [T + 1][players][34] =
columns 17..50 of GameInst::curObs after Start() and after every Step;  done [T]."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

from simlib import PortSim, RefSim, _ptr  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState, default_arena, yaw_rot  # noqa: E402

BIG = [(3584.0, 0.0), (-3584.0, 0.0), (3072.0, 4096.0), (-3072.0, -4096.0)]
SMALL = [(0.0, -4240.0), (1024.0, 0.0), (-1788.0, 2300.0), (0.0, 2816.0)]
CASES = {"1v1": (1, 3, 14), "2v2": (2, 4, 18), "3v3": (3, 3, 14)}   # name: (team, no_touch_steps, steps)


def main():
    port = PortSim(); verts, tris = port.procedural_mesh()
    ref = RefSim(verts, tris)
    out = {"mesh_verts": verts, "mesh_tris": tris, "names": np.array(list(CASES))}
    ref.lib.ref_list_setter_then_random(1)
    try:
        for name, (team, nts, T) in CASES.items():
            nc = 2 * team
            s = default_arena(nc)
            spots = [BIG[0], SMALL[0], BIG[2], SMALL[2], BIG[1], SMALL[3]]
            for c in range(nc):
                s.cars[c].pos[:] = (spots[c][0], spots[c][1], 17.0); s.cars[c].rot[:] = yaw_rot(0.3 * c); s.cars[c].boost = 5.0
            s.ball.pos[:] = (500.0, 900.0, 600.0)
            arr = (ArenaState * 1)(s)
            actions = np.zeros((T, nc), np.int32)     # action 0 every step: nobody drives off (the pads are taken in the first tick)
            D = 51 + 19 * nc
            cur = np.zeros((T + 1, nc, D), np.float32); stp = np.zeros((T, nc, D), np.float32)
            rew = np.zeros((T, nc), np.float32); done = np.zeros(T, np.int32); tr = np.zeros((T, 6), np.float32)
            resets = np.zeros(T + 1, np.int32); order = np.zeros((T + 1, nc), np.int32)
            d = ref.lib.ref_gameinst_run(team, 8, 0, 0, nts, arr, 1, _ptr(actions), T, _ptr(cur), _ptr(stp), _ptr(rew), _ptr(done), _ptr(tr), _ptr(resets), _ptr(order))
            assert d == D
            pads = cur[:, :, 17:51].copy()
            ends = np.flatnonzero(done)
            assert len(ends) >= 2, (name, done)
            first = int(ends[0])
            assert (pads[0] == 1).all(), "the chosen start state has every pad active"
            assert (pads[first] == 0).sum() >= nc, f"{name}: every car emptied its pad during the first episode"
            assert (pads[first + 1] == 1).all(), f"{name}: the reference's RandomState shows all pads active in the new episode's first observation"
            print(f"{name}: first episode ends at step {first}; inactive pad entries before / after the reset: {(pads[first] == 0).sum()} / {(pads[first + 1] == 0).sum()}")
            out[f"pr/{name}/cfg"] = np.array([team, 8, nts], np.int32)
            out[f"pr/{name}/start"] = np.frombuffer(bytes(s), np.uint8).copy()
            out[f"pr/{name}/actions"] = actions; out[f"pr/{name}/pads"] = pads; out[f"pr/{name}/done"] = done
    finally:
        ref.lib.ref_list_setter_then_random(0)
    np.savez_compressed(os.path.join(HERE, "padreset_golden.npz"), **out)
    print("wrote padreset_golden.npz")


if __name__ == "__main__":
    main()
