"""Adds gym/<case>/start_raw to tests/golden/sim_golden.npz: the state the reference Gym was RESET TO (make_sim_golden.py:gym_cases /
one_team_cases build it), next to gym/<case>/start = that state read back from the arena (once through Bullet units and back).  The
reference's rollout continued from its internal state, i.e. from start_raw; a replay from the read-back copy starts one rounding away (the
physics tapes have had phys/<name>/start_raw for the same reason).  The car order of the recording arena is taken from `start`.  Needs the
reference only for its action table (gym_cases picks action indices from it).   python tests/golden/add_gym_start_raw.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)
from simlib import PortSim, RefSim  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState  # noqa: E402
import make_sim_golden as M  # noqa: E402

ref = None
for fname, maker in (("sim_golden.npz", M.gym_cases), ("sim_golden_one_team.npz", M.one_team_cases)):
    path = os.path.join(HERE, fname)
    gold = dict(np.load(path))
    if ref is None:
        ref = RefSim(gold["mesh_verts"], gold["mesh_tris"])
    n = 0
    for case, team, tick_skip, omp, rk, nts, s0, acts in maker(ref):
        key = f"gym/{case}/start"
        if key not in gold:
            continue
        s = ArenaState.from_buffer_copy(bytes(s0))
        s.car_order = ArenaState.from_buffer_copy(gold[key].tobytes()).car_order
        gold[f"gym/{case}/start_raw"] = np.frombuffer(bytes(s), np.uint8).copy()
        n += 1
    np.savez_compressed(path, **gold)
    print(fname, ": added start_raw to", n, "gym cases")
