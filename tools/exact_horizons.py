"""How long each recorded reference tape stays bit-identical on the host build of the stepper (development tool; prints the first recorded
tick at which a tape is no longer equal to the reference's, and the one-tick pairs that are not bit-equal, per scenario).
    python tools/exact_horizons.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from rlgymppo_cpp_amd.state import ArenaState  # noqa: E402
from simlib import PortSim, state_vec  # noqa: E402

sg = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
port = PortSim(); port.set_mesh(sg["mesh_verts"], sg["mesh_tris"])
port.lib.port_run_tape.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
every = int(sg["phys_every"])
for name in [str(x) for x in sg["phys_names"]]:
    st = ArenaState.from_buffer_copy(sg[f"phys/{name}/start_raw"].tobytes())
    tape = np.ascontiguousarray(sg[f"phys/{name}/tape"], np.float32); want = sg[f"phys/{name}/states"]
    outs = (ArenaState * (len(tape) // every))()
    port.lib.port_run_tape(C.byref(st), tape.ctypes.data, len(tape), every, C.byref(outs))
    first = None
    for j in range(len(tape) // every):
        if not np.array_equal(state_vec(outs[j]), want[j]):
            first = (j + 1) * every; break
    if first is not None:
        print(f"{name}: leaves the reference in ({first - every}, {first}] of {len(tape)}")
ss = np.load(os.path.join(ROOT, "tests", "golden", "sim_steps.npz"))
names = [str(x) for x in ss["phys_names"]]
bad = {}
tot = 0
for nc in (2, 4, 6):
    B, A, T = ss[f"nc{nc}/before"], ss[f"nc{nc}/after"], ss[f"nc{nc}/tag"]
    for i in range(len(B)):
        s = ArenaState.from_buffer_copy(B[i].tobytes()); port.step(s, 1)
        tot += 1
        if not np.array_equal(state_vec(s), state_vec(ArenaState.from_buffer_copy(A[i].tobytes()))):
            bad.setdefault(names[T[i][0]], []).append(int(T[i][1]))
print("one-tick pairs:", tot, "not bit-equal:", sum(len(v) for v in bad.values()))
for k, v in bad.items(): print("  ", k, v)
