"""Golden vectors for the triangle visiting order of the arena mesh: what the REAL reference (oracle/_ref/libref_oracle.so, built from
/root/reference by oracle/Makefile) reports through btBvhTriangleMeshShape::processAllTriangles over an all-enclosing box
(oracle/ref_driver.cpp:ref_mesh_visit_order), for (a) the procedural soccar mesh and (b) a clustered triangle soup with flat and
axis-aligned triangles, 1500 of them (deep enough for the subtree-header order of btQuantizedBvh to show).  RocketSim initialises once
per process, so every mesh is run in a child process.

    python tests/golden/make_mesh_order_golden.py        ->  tests/golden/mesh_order_golden.npz
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def soup(seed, n_tris):
    rng = np.random.default_rng(seed)
    k = rng.integers(2, 6)
    centers = rng.uniform(-4000, 4000, (k, 3)); centers[:, 2] = rng.uniform(0, 2000, k)
    which = rng.integers(0, k, n_tris); scale = rng.choice([50, 300, 1500], n_tris, p=[.5, .3, .2])
    base = centers[which] + rng.normal(0, 1, (n_tris, 3)) * scale[:, None]
    tri = base[:, None, :] + rng.uniform(-150, 150, (n_tris, 3, 3))
    flat = rng.random(n_tris) < 0.2; ax = rng.integers(0, 3, n_tris)
    for i in np.nonzero(flat)[0]:
        tri[i, :, ax[i]] = tri[i, 0, ax[i]]
    return tri.reshape(-1, 3).astype(np.float32), np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3)


def ref_order(verts, tris):
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so"))
    verts = np.ascontiguousarray(verts, np.float32); tris = np.ascontiguousarray(tris, np.int32)
    assert ref.ref_init(verts.ctypes.data_as(C.c_void_p), len(verts), tris.ctypes.data_as(C.c_void_p), len(tris)) == 0
    out = np.zeros(len(tris), np.int32)
    assert ref.ref_mesh_visit_order(out.ctypes.data_as(C.c_void_p), len(tris)) == len(tris)
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        d = np.load(sys.argv[2])
        np.save(sys.argv[3], ref_order(d["verts"], d["tris"]))
        return
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    from simlib import PortSim
    pv, pt = PortSim().procedural_mesh()
    sv, st = soup(20261003, 1500)
    out = {"soup/verts": sv, "soup/tris": st}
    for name, (v, t) in {"procedural": (pv, pt), "soup": (sv, st)}.items():
        np.savez("/tmp/_mesh_in.npz", verts=np.asarray(v, np.float32).reshape(-1, 3), tris=np.asarray(t, np.int32).reshape(-1, 3))
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", "/tmp/_mesh_in.npz", "/tmp/_mesh_out.npy"])
        out[f"{name}/order"] = np.load("/tmp/_mesh_out.npy")
    np.savez_compressed(os.path.join(HERE, "mesh_order_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
