"""Generates tests/golden/seam_golden.npz from the REAL reference with a mesh of SEVERAL FILES: the procedural arena split into two
.cmf files (triangles with centroid x < 0 / x >= 0), loaded by the reference through its own per-file path -- one btBvhTriangleMeshShape,
one static rigid body and therefore one contact manifold per file (RS/RocketSim.cpp:102-212, RS/Sim/Arena/Arena.cpp:1028-1054) -- via
oracle/ref_driver.cpp:ref_init_dir.  The seam runs through the panel above each goal mouth and through the goal roofs (one quad = two
triangles, one in each file), so a ball or a car hitting them near x = 0 touches two mesh objects at once.

    python tests/golden/make_seam_golden.py          (build container; a process of its own: the reference initialises once)

Contents (data only): the two files' triangles (mesh_verts, mesh_tris, mesh_parts), and per scenario the start state, the control tape,
the reference's states every 10 ticks and one-tick pairs, as in make_sim_golden.py.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, HERE)

from simlib import PortSim, RefSim, state_vec  # noqa: E402
from rlgymppo_cpp_amd.state import ArenaState, default_arena, yaw_rot, euler_rot  # noqa: E402

Z = [0.0] * 8


def split_mesh(verts, tris):
    """Two files: centroid x < 0, centroid x >= 0 -- in that (name) order; returns the reordered triangles and the per-file counts."""
    cen = verts[tris].mean(axis=1)
    a = tris[cen[:, 0] < 0]; b = tris[cen[:, 0] >= 0]
    return np.concatenate([a, b]).astype(np.int32), [len(a), len(b)]


def write_files(verts, tris, parts, root):
    from simlib import write_cmf_parts
    return write_cmf_parts(verts, tris, parts, root)


def scenarios():
    out = {}
    # ball into the panel above the goal mouth, on the seam (the quad's diagonal), with spin; then it drops along the wall
    s = default_arena(2); s.ball.pos[:] = (20, 4300, 1100); s.ball.vel[:] = (-60, 2600, 150); s.ball.ang_vel[:] = (2, -1, 3)
    out["ball_panel_seam"] = (s, lambda t, k: Z, 300)
    # ball into the goal, up against the roof near x = 0, back wall of the goal
    s = default_arena(2); s.ball.pos[:] = (-30, 4800, 300); s.ball.vel[:] = (40, 1500, 1400)
    out["ball_goal_roof_seam"] = (s, lambda t, k: Z, 300)
    # car nose first into the panel above the goal at x = 0: the hitbox is deep in both triangles (two manifolds, EPA)
    s = default_arena(2); c = s.cars[0]; c.pos[:] = (5, 4300, 1200); c.rot[:] = yaw_rot(np.pi / 2); c.vel[:] = (0, 1800, 0); c.flags = 0; c.boost = 100
    out["car_into_panel_seam"] = (s, lambda t, k: [1, 0, 0, 0, 0, 0, 1, 0] if k == 0 else Z, 240)
    # car tumbling into the goal roof across the seam
    s = default_arena(2); c = s.cars[0]; c.pos[:] = (-60, 5300, 300); c.rot[:] = euler_rot(0.4, 1.2, 0.3); c.vel[:] = (150, 200, 1500); c.ang_vel[:] = (1.5, -2.0, 0.7); c.flags = 0
    out["car_goal_roof_seam"] = (s, lambda t, k: [1, 0.3, 0.5, 0, -1, 0, 0, 0] if k == 0 else Z, 300)
    # car driving along the back wall's foot through x = 0 inside the goal mouth region is floor only; instead: sliding along the panel
    s = default_arena(2); c = s.cars[0]; c.pos[:] = (-700, 5050, 1300); c.rot[:] = euler_rot(0.0, 0.0, np.pi / 2); c.vel[:] = (1400, 250, 0); c.flags = 0; c.boost = 60
    out["car_slides_along_panel"] = (s, lambda t, k: [1, 0, 0, 0.2, 0, 0, 1 if t < 60 else 0, 0] if k == 0 else Z, 300)
    return out


def main():
    port = PortSim()
    verts, tris = port.procedural_mesh()
    tris, parts = split_mesh(verts, tris)
    root = write_files(verts, tris, parts, tempfile.mkdtemp(prefix="seam_mesh_"))
    ref = RefSim(None, None, mesh_dir=root)
    out = {"mesh_verts": verts, "mesh_tris": tris, "mesh_parts": np.array(parts, np.int32)}
    every = 10; names = []
    before_l, after_l, tag_l = [], [], []
    mbuf = np.zeros((64, 16), np.float32)
    two = 0
    for si, (name, (s0, fn, ticks)) in enumerate(scenarios().items()):
        nc = s0.num_cars
        a = ref.arena(nc // 2); ref.set_state(a, s0)
        s0.car_order = ref.get_state(a).car_order
        pair_arena = ref.arena(nc // 2)
        tape = np.zeros((ticks, nc, 8), np.float32); rec = []
        for t in range(ticks):
            for k in range(nc):
                tape[t, k] = fn(t, k); ref.set_controls(a, k, list(tape[t, k]))
            before = ref.get_state(a)
            ref.step(a, 1)
            after = ref.get_state(a)
            nman = ref.lib.ref_debug_manifolds(a, mbuf.ctypes.data_as(C.c_void_p), 64)
            if nman > 0:
                mans = {int(mbuf[q][2]) for q in range(nman) if int(mbuf[q][1]) == -1}
                two += len(mans) >= 2
                ref.set_state(pair_arena, before)
                before.car_order = ref.get_state(pair_arena).car_order
                for k in range(nc): ref.set_controls(pair_arena, k, list(tape[t, k]))
                ref.step(pair_arena, 1)
                before_l.append(np.frombuffer(bytes(before), np.uint8).copy()); after_l.append(np.frombuffer(bytes(ref.get_state(pair_arena)), np.uint8).copy()); tag_l.append((si, t))
            if (t + 1) % every == 0:
                rec.append(state_vec(after))
        out[f"phys/{name}/start_raw"] = np.frombuffer(bytes(s0), np.uint8).copy()
        out[f"phys/{name}/tape"] = tape; out[f"phys/{name}/states"] = np.stack(rec)
        names.append(name)
    out["phys_names"] = np.array(names); out["phys_every"] = np.int32(every)
    out["pairs/before"] = np.stack(before_l); out["pairs/after"] = np.stack(after_l); out["pairs/tag"] = np.array(tag_l, np.int32)
    np.savez_compressed(os.path.join(HERE, "seam_golden.npz"), **out)
    print("wrote seam_golden.npz:", len(names), "scenarios,", len(before_l), "one-tick pairs; ticks with world manifolds of >= 2 static bodies:", two, "; files", parts)


if __name__ == "__main__":
    main()
