"""Golden vectors for the two narrowphase routines the stepper restates from Bullet, recorded from the REAL reference's own code
(oracle/_ref/libref_oracle.so: ref_gjk_box_triangle = btGjkPairDetector as btConvexConvexAlgorithm sets it up for a hitbox against a
mesh triangle; ref_adjust_internal_edge = btAdjustInternalEdgeContacts on the arena mesh's btTriangleInfoMap):

  gjk/*   3000 random Octane hitbox poses x triangles (faces, edges, vertices around the contact threshold): inputs + the detector's output
  cast/*  3000 wheel rays against a hitbox / the ball: inputs + what btCollisionWorld::rayTestSingle (btSubsimplexConvexCast) reports
  edge/*  40 contact points per triangle of the procedural arena (near its edges, normals around the face's and the neighbour's):
          inputs + the adjusted point

    python tests/golden/make_narrowphase_golden.py        ->  tests/golden/narrowphase_golden.npz
The generators of the cases are tools/gjk_fuzz.py:case and the loop below (same as tools/edge_fuzz.py).
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
FP = C.POINTER(C.c_float)


def p(a):
    return a.ctypes.data_as(FP)


def main():
    import gjk_fuzz
    from simlib import PortSim
    ref = gjk_fuzz.ref
    rng = np.random.default_rng(20261003)
    n = 3000
    pos = np.zeros((n, 3), np.float32); rot = np.zeros((n, 9), np.float32); tri = np.zeros((n, 9), np.float32)
    out = np.zeros((n, 8), np.float32); hit = np.zeros(n, np.int32)
    for i in range(n):
        pos[i], rot[i], tri[i] = gjk_fuzz.case(rng)
        hit[i] = ref.ref_gjk_box_triangle(p(gjk_fuzz.HALF), p(pos[i]), p(rot[i]), p(tri[i]), 0.0, gjk_fuzz.CBT_CAR, p(out[i]))
    res = {"gjk/half": gjk_fuzz.HALF, "gjk/breaking": np.float32(gjk_fuzz.CBT_CAR), "gjk/pos": pos, "gjk/rot": rot, "gjk/tri": tri, "gjk/hit": hit, "gjk/out": out}

    ps = PortSim(); verts, tris = ps.procedural_mesh(); ps.set_mesh(verts, tris)
    v = np.ascontiguousarray(verts, np.float32); t = np.ascontiguousarray(tris, np.int32)
    assert ref.ref_init(v.ctypes.data_as(C.c_void_p), len(v), t.ctypes.data_as(C.c_void_p), len(t)) == 0
    ref.ref_adjust_internal_edge.argtypes = [C.c_int, FP, FP, FP, C.c_float, FP]
    per = 40
    e_tri = np.zeros((len(t) * per,), np.int32); e_pb = np.zeros((len(t) * per, 3), np.float32); e_n = np.zeros((len(t) * per, 3), np.float32)
    e_d = np.zeros(len(t) * per, np.float32); e_out = np.zeros((len(t) * per, 7), np.float32)
    k = 0
    for ti in range(len(t)):
        V = v[t[ti]].astype(np.float64) * 0.02          # BT units, as the reference holds them (verts_uu * UU_TO_BT in float: see below)
        V32 = (v[t[ti]] * np.float32(0.02)).astype(np.float32)
        fn = np.cross(V[1] - V[0], V[2] - V[0]); fn /= np.linalg.norm(fn) + 1e-30
        for _ in range(per):
            e = rng.integers(0, 3); a, b = V[e], V[(e + 1) % 3]
            u = rng.choice([rng.uniform(0, 1), 0.0, 1.0, rng.uniform(0, 0.02)])
            inward = np.cross(fn, b - a); inward /= np.linalg.norm(inward) + 1e-30
            pt = a + (b - a) * u + inward * rng.choice([0.0, rng.uniform(0, 0.12), rng.uniform(0, 0.01)])
            axis = (b - a) / (np.linalg.norm(b - a) + 1e-30)
            th = rng.choice([0.0, rng.uniform(-1.6, 1.6), rng.uniform(-0.05, 0.05), rng.uniform(-3.1, 3.1)])
            nrm = fn * np.cos(th) + np.cross(axis, fn) * np.sin(th) + rng.normal(size=3) * rng.choice([0.0, 0.02])
            nrm /= np.linalg.norm(nrm)
            e_tri[k] = ti; e_pb[k] = pt; e_n[k] = nrm; e_d[k] = rng.uniform(-0.03, 0.04)
            assert ref.ref_adjust_internal_edge(ti, p(V32.reshape(9).copy()), p(e_pb[k]), p(e_n[k]), e_d[k], p(e_out[k])) == 0
            k += 1
    res.update({"edge/tri": e_tri, "edge/pb": e_pb, "edge/n": e_n, "edge/dist": e_d, "edge/out": e_out})
    # cast/*: 3000 wheel-sized rays against an Octane hitbox at a random pose or the ball: btCollisionWorld::rayTestSingle (btSubsimplexConvexCast)
    import cast_fuzz
    m = 3000
    c_from = np.zeros((m, 3), np.float32); c_to = np.zeros((m, 3), np.float32); c_rad = np.zeros(m, np.float32); c_pos = np.zeros((m, 3), np.float32)
    c_rot = np.zeros((m, 9), np.float32); c_out = np.zeros((m, 4), np.float32); c_hit = np.zeros(m, np.int32)
    for i in range(m):
        c_from[i], c_to[i], c_rad[i], c_pos[i], c_rot[i] = cast_fuzz.case(rng)
        c_hit[i] = cast_fuzz.ref.ref_ray_convex(p(c_from[i]), p(c_to[i]), p(cast_fuzz.HALF), c_rad[i], p(c_pos[i]), p(c_rot[i]), p(c_out[i]))
    res.update({"cast/half": cast_fuzz.HALF, "cast/from": c_from, "cast/to": c_to, "cast/radius": c_rad, "cast/pos": c_pos, "cast/rot": c_rot, "cast/hit": c_hit, "cast/out": c_out})
    np.savez_compressed(os.path.join(HERE, "narrowphase_golden.npz"), **res)
    print({k: v.shape for k, v in res.items()}, "gjk hits", int(hit.sum()))


if __name__ == "__main__":
    main()
