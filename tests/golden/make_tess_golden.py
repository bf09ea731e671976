"""Generates tests/golden/tess_golden.npz from the REAL reference on the TESSELLATED arena in 16 files: the procedural soccar arena at the
game meshes' density (fillets of 8 strips, no edge longer than 700 uu: 10 084 triangles) split by sector into 16 .cmf files, loaded by the
reference through its own per-file path -- one btBvhTriangleMeshShape, one static body, one contact manifold per file (RS/RocketSim.cpp:102-212,
RS/Sim/Arena/Arena.cpp:1028-1054) -- via oracle/ref_driver.cpp:ref_init_dir.  This is the mesh of bench.py's `mesh_tessellated` leg and the
only workload that reaches the stepper's device-only mesh machinery at size (BVH top levels in LDS, a frontier of up to 128 nodes, the kept
candidate leaves, up to 24 leaves per body, two mesh manifolds per body).

    python tests/golden/make_tess_golden.py          (build container; a process of its own: the reference initialises once)

Contents (data only): mesh_verts / mesh_tris (in file order) / mesh_parts (triangles per file); per tape -- kickoffs of 1v1 / 2v2 / 3v3 with
every car on random controls held for random spans, two tapes per team size, 1 200 ticks -- the start state, the control tape, the
reference's states every 10 ticks; and one-tick pairs (state in -> one tick -> state out, recorded by a second arena stepped from the
recorded state) from ticks with a mesh contact.
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

TICKS, EVERY, N_FILES, PAIRS_PER_TAPE = 1200, 10, 16, 60


def sectors(verts, tris, n_files):
    """the triangles grouped by the angle of their centroid around the field centre (rlgymppo_cpp_amd.env.write_cmf_files' split), in file order"""
    cen = verts[tris].mean(axis=1)
    sec = ((np.arctan2(cen[:, 1], cen[:, 0]) + np.pi) / (2 * np.pi) * n_files).astype(np.int64).clip(0, n_files - 1)
    order = np.argsort(sec, kind="stable")
    parts = [int((sec == k).sum()) for k in range(n_files)]
    assert all(p > 0 for p in parts)
    return tris[order].astype(np.int32), parts


def main():
    verts, tris = procedural_mesh_ex(8, 700.0)
    tris, parts = sectors(verts, tris, N_FILES)
    root = write_cmf_parts(verts, tris, parts, tempfile.mkdtemp(prefix="tess_mesh_"))
    ref = RefSim(None, None, mesh_dir=root)
    ref.lib.ref_arena_reset_kickoff.argtypes = [C.c_void_p, C.c_int]; ref.lib.ref_arena_free.argtypes = [C.c_void_p]
    out = {"mesh_verts": verts, "mesh_tris": tris, "mesh_parts": np.array(parts, np.int32)}
    names, before_l, after_l, tag_l = [], [], [], []
    mbuf = np.zeros((64, 16), np.float32)
    for ti, (team, seed) in enumerate([(1, 11), (1, 12), (2, 21), (2, 22), (3, 31), (3, 32)]):
        nc = 2 * team; rng = np.random.RandomState(seed)
        k0 = ref.arena(team); ref.lib.ref_arena_reset_kickoff(k0, seed); s0 = ref.get_state(k0); ref.lib.ref_arena_free(k0)
        a = ref.arena(team); ref.set_state(a, s0); s0.car_order = ref.get_state(a).car_order
        pair_arena = ref.arena(team)
        tape = np.zeros((TICKS, nc, 8), np.float32)
        for k in range(nc):
            t = 0
            while t < TICKS:
                span = int(rng.randint(4, 60))
                c = np.zeros(8, np.float32)
                c[0] = rng.choice([1.0, 1.0, 1.0, -1.0, 0.0]); c[1:5] = rng.choice([-1.0, 0.0, 0.0, 1.0], size=4)
                c[5] = float(rng.rand() < 0.15); c[6] = float(rng.rand() < 0.6); c[7] = float(rng.rand() < 0.1)
                tape[t:t + span, k] = c; t += span
        rec, cand = [], []
        for t in range(TICKS):
            for k in range(nc): ref.set_controls(a, k, list(tape[t, k]))
            before = ref.get_state(a)
            ref.step(a, 1)
            nman = ref.lib.ref_debug_manifolds(a, mbuf.ctypes.data_as(C.c_void_p), 64)
            world = {int(mbuf[q][2]) for q in range(nman) if int(mbuf[q][1]) == -1}
            if world: cand.append((t, before, len(world)))
            if (t + 1) % EVERY == 0: rec.append(state_vec(ref.get_state(a)))
        # pairs: every tick that touched two mesh objects at once, then an even sample of the rest
        multi = [c for c in cand if c[2] >= 2]; single = [c for c in cand if c[2] < 2]
        keep = multi[:PAIRS_PER_TAPE // 2] + [single[i] for i in np.linspace(0, len(single) - 1, min(len(single), PAIRS_PER_TAPE - min(len(multi), PAIRS_PER_TAPE // 2))).astype(int)] if single else multi[:PAIRS_PER_TAPE]
        for (t, before, nw) in sorted(keep, key=lambda c: c[0]):
            ref.set_state(pair_arena, before); before.car_order = ref.get_state(pair_arena).car_order
            for k in range(nc): ref.set_controls(pair_arena, k, list(tape[t, k]))
            ref.step(pair_arena, 1)
            before_l.append(np.frombuffer(bytes(before), np.uint8).copy()); after_l.append(np.frombuffer(bytes(ref.get_state(pair_arena)), np.uint8).copy()); tag_l.append((ti, t, nw))
        name = f"{team}v{team}_seed{seed}"
        out[f"phys/{name}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()
        out[f"phys/{name}/tape"] = tape; out[f"phys/{name}/states"] = np.stack(rec)
        names.append(name)
        print(f"{name}: {len(cand)} ticks with a mesh contact ({len(multi)} on two or more files at once), {len(keep)} pairs kept", flush=True)
        ref.lib.ref_arena_free(a); ref.lib.ref_arena_free(pair_arena)
    out["phys_names"] = np.array(names); out["phys_every"] = np.int32(EVERY)
    out["pairs/before"] = np.stack(before_l); out["pairs/after"] = np.stack(after_l); out["pairs/tag"] = np.array(tag_l, np.int32)
    path = os.path.join(HERE, "tess_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes:", len(names), "tapes,", len(before_l), "one-tick pairs,", len(tris), "triangles in", len(parts), "files")


if __name__ == "__main__":
    main()
