"""Fuzz of the host build of csrc/arena_gjk.h:gjk_box_triangle against the reference's own btGjkPairDetector (oracle/ref_driver.cpp:
ref_gjk_box_triangle): random Octane hitbox poses against random triangles placed so that the closest features are faces, edges and
vertices at distances around the contact threshold.  Development tool (build container; needs oracle/_ref).

    python tools/gjk_fuzz.py [n_cases] [seed]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
port = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_port.so"))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so"))
FP = C.POINTER(C.c_float)
ref.ref_gjk_box_triangle.argtypes = [FP, FP, FP, FP, C.c_float, C.c_float, FP]
port.port_gjk_box_triangle.argtypes = [FP, FP, FP, C.c_float, FP]
HALF = (np.array([120.507, 86.6994, 38.6591], np.float32) * np.float32(0.02)) / np.float32(2)     # K::HITBOX_* * UU2BT / 2
CBT_CAR = np.float32(0.040624548)


def p(a):
    return a.ctypes.data_as(FP)


def rand_rot(rng):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float32)


def case(rng):
    R = rand_rot(rng) if rng.random() < 0.8 else np.eye(3, dtype=np.float32)
    pos = rng.uniform(-40, 40, 3).astype(np.float32)
    # a point on (or near) the box surface, a triangle through a point at signed distance d from it along a roughly outward direction
    s = rng.uniform(-1, 1, 3); k = rng.integers(0, 3); s[k] = np.sign(s[k]) if s[k] != 0 else 1.0
    if rng.random() < 0.3: s[(k + 1) % 3] = np.sign(s[(k + 1) % 3]) or 1.0          # an edge
    if rng.random() < 0.3: s[(k + 2) % 3] = np.sign(s[(k + 2) % 3]) or 1.0          # ... or a corner
    surf = R @ (s * HALF) + pos
    out = R @ (np.sign(s) * (np.abs(s) > 0.999)); out = out / (np.linalg.norm(out) + 1e-9)
    d = rng.choice([rng.uniform(-0.03, 0.0), rng.uniform(0.0, 0.05), rng.uniform(0.03, 0.06)])
    base = surf + out * d
    size = rng.choice([0.5, 3.0, 20.0])
    mode = rng.random()
    if mode < 0.5:      # a face roughly facing the box
        t1 = np.cross(out, rng.normal(size=3)); t1 /= np.linalg.norm(t1) + 1e-9; t2 = np.cross(out, t1)
        tilt = rng.normal(size=3) * rng.choice([0.0, 0.02, 0.3])
        v = [base + (t1 * a + t2 * b) * size + tilt * (a + b) for a, b in ((-1, -0.6), (1, -0.6), (0, 1.2))]
    elif mode < 0.8:    # an edge passing through the point
        e = rng.normal(size=3); e /= np.linalg.norm(e)
        v = [base - e * size, base + e * size, base + out * size * rng.uniform(0.2, 1.0) + rng.normal(size=3) * size * 0.5]
    else:               # a vertex at the point
        v = [base, base + (out + rng.normal(size=3) * 0.7) * size, base + (out + rng.normal(size=3) * 0.7) * size]
    tri = np.array(v, np.float32).reshape(9)
    return pos, R.reshape(9).copy(), tri


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    stats = {"both": 0, "none": 0, "flag": 0, "deep": 0}; worst = np.zeros(3); bad = []
    for i in range(n):
        pos, R, tri = case(rng)
        a = np.zeros(8, np.float32); b = np.zeros(8, np.float32)
        ha = ref.ref_gjk_box_triangle(p(HALF), p(pos), p(R), p(tri), 0.0, CBT_CAR, p(a))
        hb = port.port_gjk_box_triangle(p(pos), p(R), p(tri), CBT_CAR, p(b))
        if b[7] != 0:                                   # the penetration-depth arena was too small (never on the host build)
            stats["deep"] += 1; continue
        if ha != hb:
            stats["flag"] += 1; bad.append((i, "flag", ha, hb, a[6], b[6])); continue
        if not ha:
            stats["none"] += 1; continue
        stats["both"] += 1
        e = np.array([np.abs(a[0:3] - b[0:3]).max(), np.abs(a[3:6] - b[3:6]).max(), abs(a[6] - b[6])])
        worst = np.maximum(worst, e)
        if e[0] > 1e-5 or e[1] > 1e-4 or e[2] > 1e-5: bad.append((i, "value", *e))
    print(stats, "worst |dn| %.3g |dp| %.3g |dd| %.3g" % tuple(worst), "box margin", a[7])
    st = (C.c_int * 64)(); port.port_epa_stats(st, 1)
    print("EPA runs", st[0], "max support vertices", st[1], "max faces", st[2], "max iterations", st[3], "status counts", list(st[4:15]), "\n  vertices / 4 histogram", list(st[16:48]))
    for x in bad[:20]: print("  ", x)
    print(len(bad), "cases beyond 1e-5 / 1e-4 / 1e-5")


if __name__ == "__main__":
    main()
