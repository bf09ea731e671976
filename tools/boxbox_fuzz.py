"""box_box_ode (csrc/arena_world.h) against the reference's btBoxBoxDetector on random pairs of Octane hitboxes in touching poses: the number of
points, and every point's normal / position / depth, compared for equality (development tool; needs oracle/_ref).   usage: boxbox_fuzz.py [n] [seed]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from simlib import PortSim, RefSim
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
port = PortSim().lib; ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so"))
half = np.zeros(3, np.float32); port.port_hitbox_ctor_half(C.c_void_p(half.ctypes.data))     # what btBoxShape's constructor is given (BT units)
P = C.c_void_p
def rot():
    q = rng.randn(4); q /= np.linalg.norm(q); w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float32)
bad = 0; hits = 0; multi = 0
for it in range(n):
    R1, R2 = np.ascontiguousarray(rot()), np.ascontiguousarray(rot())
    p1 = (rng.randn(3) * 2).astype(np.float32)
    d = rng.randn(3); d /= np.linalg.norm(d)
    p2 = (p1 + d * rng.uniform(0.3, 2.6)).astype(np.float32)
    o_r = np.zeros((8, 7), np.float32); o_p = np.zeros((8, 7), np.float32)
    nr = ref.ref_box_box(P(half.ctypes.data), P(p1.ctypes.data), P(R1.ctypes.data), P(p2.ctypes.data), P(R2.ctypes.data), P(o_r.ctypes.data), 8)
    npt = port.port_box_box(P(p1.ctypes.data), P(R1.ctypes.data), P(p2.ctypes.data), P(R2.ctypes.data), P(o_p.ctypes.data))
    hits += nr > 0; multi += nr > 1
    if nr != npt or not np.array_equal(o_r[:nr].view(np.uint32), o_p[:npt].view(np.uint32)):
        bad += 1
        if bad <= 3:
            print("case", it, "reference", nr, "points, port", npt); print(o_r[:nr]); print(o_p[:npt])
print(f"{n} pairs, {hits} touching ({multi} with several points): {bad} differ")
