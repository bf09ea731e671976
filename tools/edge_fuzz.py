"""Fuzz of the host build of csrc/arena_world.h:adjust_internal_edge against the reference's btAdjustInternalEdgeContacts
(oracle/ref_driver.cpp:ref_adjust_internal_edge) on the procedural arena mesh: contact points near the edges and corners of every
triangle, with normals around the face normal and around the neighbour's.  Development tool (build container; needs oracle/_ref).

    python tools/edge_fuzz.py [cases_per_triangle] [seed]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simlib import PortSim  # noqa: E402

FP = C.POINTER(C.c_float)


def p(a):
    return a.ctypes.data_as(FP)


def main():
    per = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ps = PortSim(); verts, tris = ps.procedural_mesh(); ps.set_mesh(verts, tris)
    port = ps.lib
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so"))
    v = np.ascontiguousarray(verts, np.float32); t = np.ascontiguousarray(tris, np.int32)
    assert ref.ref_init(v.ctypes.data_as(C.c_void_p), len(v), t.ctypes.data_as(C.c_void_p), len(t)) == 0
    ref.ref_adjust_internal_edge.argtypes = [C.c_int, FP, FP, FP, C.c_float, FP]
    port.port_adjust_internal_edge.argtypes = [C.c_int, FP, FP, C.c_float, FP]
    port.port_mesh_triangle.argtypes = [C.c_int, FP, C.POINTER(C.c_uint32), FP]
    n_tris = len(t)
    worst = np.zeros(3); n_cases = 0; changed = 0; bad = []
    for si in range(n_tris):
        tri = np.zeros(9, np.float32); fl = C.c_uint32(); ang = np.zeros(3, np.float32)
        src = port.port_mesh_triangle(si, p(tri), C.byref(fl), p(ang))
        V = tri.reshape(3, 3).astype(np.float64)
        fn = np.cross(V[1] - V[0], V[2] - V[0]); fn /= np.linalg.norm(fn) + 1e-30
        for _ in range(per):
            e = rng.integers(0, 3); a, b = V[e], V[(e + 1) % 3]
            u = rng.choice([rng.uniform(0, 1), 0.0, 1.0, rng.uniform(0, 0.02)])
            inward = np.cross(fn, b - a); inward /= np.linalg.norm(inward) + 1e-30
            pt = a + (b - a) * u + inward * rng.choice([0.0, rng.uniform(0, 0.12), rng.uniform(0, 0.01)])
            axis = (b - a) / (np.linalg.norm(b - a) + 1e-30)
            th = rng.choice([0.0, rng.uniform(-1.6, 1.6), rng.uniform(-0.05, 0.05), rng.uniform(-3.1, 3.1)])
            nrm = fn * np.cos(th) + np.cross(axis, fn) * np.sin(th) + rng.normal(size=3) * rng.choice([0.0, 0.02])
            if rng.random() < 0.8: nrm /= np.linalg.norm(nrm)
            else: nrm *= rng.uniform(0.9, 1.1)          # (GJK's normal is normalised, but edges 1 and 2 of the reference use it raw)
            dist = np.float32(rng.uniform(-0.03, 0.04))
            pb = pt.astype(np.float32); n3 = nrm.astype(np.float32)
            a7 = np.zeros(7, np.float32); b7 = np.zeros(7, np.float32)
            assert ref.ref_adjust_internal_edge(src, p(tri), p(pb), p(n3), dist, p(a7)) == 0
            assert port.port_adjust_internal_edge(si, p(pb), p(n3), dist, p(b7)) == 0
            n_cases += 1
            if np.abs(a7[0:3] - n3).max() > 0: changed += 1
            err = np.array([np.abs(a7[0:3] - b7[0:3]).max(), np.abs(a7[3:6] - b7[3:6]).max(), abs(a7[6] - b7[6])])
            worst = np.maximum(worst, err)
            if err[0] > 1e-6 or err[1] > 1e-5: bad.append((si, src, e, err[0], err[1], a7[:3], b7[:3], n3))
    print(f"{n_cases} cases on {n_tris} triangles, {changed} adjusted by the reference; worst |dn| {worst[0]:.3g} |dp| {worst[1]:.3g} |dd| {worst[2]:.3g}; {len(bad)} beyond 1e-6 / 1e-5")
    for x in bad[:12]: print("  ", x)


if __name__ == "__main__":
    main()
