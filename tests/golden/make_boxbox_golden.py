"""Golden vectors for csrc/arena_world.h:box_box_ode: the reference's btBoxBoxDetector (oracle/_ref, ref_box_box) on 600 seeded pairs of
Octane hitboxes -- random relative poses, a third of them nearly aligned (face contacts with four clipped points), a few edge-edge.
Writes tests/golden/boxbox_golden.npz: pos1/rot1/pos2/rot2, ctor_half, n (points reported), pts [case][8][7] = normal, point, depth.
    PYTHONPATH=. python tests/golden/make_boxbox_golden.py        (needs /root/reference's build in oracle/_ref)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simlib import PortSim


def quat_rot(q):
    q = q / np.linalg.norm(q); w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float32)


def cases(n, seed=7):
    rng = np.random.RandomState(seed)
    for it in range(n):
        R1 = quat_rot(rng.randn(4))
        R2 = quat_rot(rng.randn(4)) if it % 3 else (R1 @ quat_rot(np.array([1.0, 0, 0, 0]) + 0.05 * rng.randn(4))).astype(np.float32)
        p1 = (rng.randn(3) * 2).astype(np.float32)
        d = rng.randn(3); d /= np.linalg.norm(d)
        p2 = (p1 + d * rng.uniform(0.3, 2.4)).astype(np.float32)
        yield p1, np.ascontiguousarray(R1, np.float32), p2, np.ascontiguousarray(R2, np.float32)


def main():
    ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so")); port = PortSim().lib
    half = np.zeros(3, np.float32); port.port_hitbox_ctor_half(C.c_void_p(half.ctypes.data))
    P = C.c_void_p
    out = {k: [] for k in ("pos1", "rot1", "pos2", "rot2", "n", "pts")}
    for p1, R1, p2, R2 in cases(600):
        o = np.zeros((8, 7), np.float32)
        n = ref.ref_box_box(P(half.ctypes.data), P(p1.ctypes.data), P(R1.ctypes.data), P(p2.ctypes.data), P(R2.ctypes.data), P(o.ctypes.data), 8)
        out["pos1"].append(p1); out["rot1"].append(R1); out["pos2"].append(p2); out["rot2"].append(R2); out["n"].append(n); out["pts"].append(o)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "boxbox_golden.npz"), ctor_half=half, **{k: np.array(v) for k, v in out.items()})
    n = np.array(out["n"]); print("600 pairs:", int((n > 0).sum()), "touching,", int((n > 1).sum()), "with several points,", int((n == 4).sum()), "with four")


if __name__ == "__main__":
    main()
