"""Generates tests/golden/wedge_golden.npz from the REAL reference: ticks whose contacts do NOT fit the stepper's LDS-resident contact layout.

The tessellated procedural arena (10 084 triangles) dealt ROUND-ROBIN into 16 .cmf files -- neighbouring triangles lie in different files, every
file's box spans the arena -- and loaded by the reference through its own per-file path (one btBvhTriangleMeshShape, one static body, one
contact manifold per file: RS/RocketSim.cpp:102-212, RS/Sim/Arena/Arena.cpp:1028-1054; oracle/ref_driver.cpp:ref_init_dir).  A car that leans
on a fillet or sits in a corner now holds points in three, four, ... manifolds at once, and every body has sixteen listed mesh manifolds for the
island sort.  On top of that: six-car pile-ups (all cars boosting at the ball from a kickoff, then at each other), for the car-pair pool.

    python tests/golden/make_wedge_golden.py          (build container; a process of its own: the reference initialises once)

Contents (data only): mesh_verts / mesh_tris (in file order) / mesh_parts; per tape the start state, the control tape, the reference's states
every 10 ticks, and per tick the largest number of static manifolds WITH points on one body and the number of car-car points.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

from simlib import RefSim, state_vec, write_cmf_parts  # noqa: E402
from rlgymppo_cpp_amd.env import procedural_mesh_ex  # noqa: E402

EVERY, N_FILES = 10, 16


def main():
    verts, tris = procedural_mesh_ex(8, 700.0)
    order = np.argsort(np.arange(len(tris)) % N_FILES, kind="stable")      # triangle i goes to file i mod 16
    parts = [int((np.arange(len(tris)) % N_FILES == k).sum()) for k in range(N_FILES)]
    tris = tris[order].astype(np.int32)
    root = write_cmf_parts(verts, tris, parts, tempfile.mkdtemp(prefix="wedge_mesh_"))
    ref = RefSim(None, None, mesh_dir=root)
    ref.lib.ref_arena_reset_kickoff.argtypes = [C.c_void_p, C.c_int]; ref.lib.ref_arena_free.argtypes = [C.c_void_p]
    out = {"mesh_verts": verts, "mesh_tris": tris, "mesh_parts": np.array(parts, np.int32)}
    names = []
    mbuf = np.zeros((256, 16), np.float32)
    # (team, seed, ticks, kind): random driving as in make_tess_golden.py; `pile`: everybody boosts at the ball, steering at random after the hit
    for (team, seed, ticks, kind) in [(1, 41, 1200, "drive"), (2, 42, 1200, "drive"), (3, 43, 1200, "drive"), (3, 44, 600, "pile"), (3, 45, 600, "pile")]:
        nc = 2 * team; rng = np.random.RandomState(seed)
        k0 = ref.arena(team); ref.lib.ref_arena_reset_kickoff(k0, seed); s0 = ref.get_state(k0); ref.lib.ref_arena_free(k0)
        a = ref.arena(team); ref.set_state(a, s0); s0.car_order = ref.get_state(a).car_order
        tape = np.zeros((ticks, nc, 8), np.float32)
        for k in range(nc):
            t = 0
            while t < ticks:
                span = int(rng.randint(4, 60))
                c = np.zeros(8, np.float32)
                if kind == "pile" and t < 140:
                    c[0] = 1.0; c[6] = 1.0; span = 140 - t
                else:
                    c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
                    c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
                tape[t:t + span, k] = c; t += span
        rec, tag = [], np.zeros((ticks, 2), np.int32)
        for t in range(ticks):
            for k in range(nc): ref.set_controls(a, k, list(tape[t, k]))
            ref.step(a, 1)
            n = ref.lib.ref_debug_manifolds(a, mbuf.ctypes.data_as(C.c_void_p), 256)
            per_body = {}
            pair_pts = 0
            for q in range(n):
                b0, b1, m = int(mbuf[q][0]), int(mbuf[q][1]), int(mbuf[q][2])
                if b0 >= 1 and b1 >= 1: pair_pts += 1            # (BodyKind: -1 static, 0 ball, car id >= 1)
                if b1 == -1: per_body.setdefault(b0, set()).add(m)
                elif b0 == -1: per_body.setdefault(b1, set()).add(m)
            tag[t] = (max([len(v) for v in per_body.values()] + [0]), pair_pts)
            if (t + 1) % EVERY == 0: rec.append(state_vec(ref.get_state(a)))
        name = f"{team}v{team}_{kind}_seed{seed}"
        out[f"phys/{name}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()
        out[f"phys/{name}/tape"] = tape; out[f"phys/{name}/states"] = np.stack(rec); out[f"phys/{name}/tag"] = tag
        names.append(name)
        print(f"{name}: ticks with >= 3 / >= 5 static manifolds with points on one body: {(tag[:, 0] >= 3).sum()} / {(tag[:, 0] >= 5).sum()} (max {tag[:, 0].max()}); "
              f"car-car points max {tag[:, 1].max()}, ticks with > 4: {(tag[:, 1] > 4).sum()}", flush=True)
        ref.lib.ref_arena_free(a)
    out["phys_names"] = np.array(names); out["phys_every"] = np.int32(EVERY)
    path = os.path.join(HERE, "wedge_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes:", len(names), "tapes,", len(tris), "triangles in", len(parts), "files")


if __name__ == "__main__":
    main()
