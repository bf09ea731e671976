"""Fuzz of the host build of csrc/arena_simplex.h:ray_convex_cast (through arena_world.h:ray_convex_hit) against the reference's own
btCollisionWorld::rayTestSingle -> btSubsimplexConvexCast (oracle/ref_driver.cpp:ref_ray_convex): wheel-sized rays against an Octane
hitbox at a random pose and against the ball.  Development tool (build container; needs oracle/_ref).

    python tools/cast_fuzz.py [n_cases] [seed]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
port = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_port.so"))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_oracle.so"))
FP = C.POINTER(C.c_float)
for lib, fn in ((ref, "ref_ray_convex"), (port, "port_ray_convex")):
    getattr(lib, fn).argtypes = [FP, FP, FP, C.c_float, FP, FP, FP]
HALF = (np.array([120.507, 86.6994, 38.6591], np.float32) * np.float32(0.02)) / np.float32(2)
BALL_R = np.float32(91.25 * 0.02)


def p(a):
    return a.ctypes.data_as(FP)


def case(rng):
    from gjk_fuzz import rand_rot
    R = rand_rot(rng) if rng.random() < 0.85 else np.eye(3, dtype=np.float32)
    pos = rng.uniform(-40, 40, 3).astype(np.float32)
    sphere = rng.random() < 0.25
    # a ray of wheel length (~0.5 .. 1.2 BT) that passes near the surface: pick a surface point, a direction roughly into the shape
    if sphere:
        d = rng.normal(size=3); d /= np.linalg.norm(d); surf = pos + d * BALL_R; out = d
    else:
        s = rng.uniform(-1, 1, 3); k = rng.integers(0, 3); s[k] = np.sign(s[k]) or 1.0
        if rng.random() < 0.3: s[(k + 1) % 3] = np.sign(s[(k + 1) % 3]) or 1.0
        surf = R @ (s * HALF) + pos
        out = R @ (np.sign(s) * (np.abs(s) > 0.999)); out = out / (np.linalg.norm(out) + 1e-9)
    dirn = -out + rng.normal(size=3) * rng.choice([0.0, 0.1, 0.6]); dirn /= np.linalg.norm(dirn) + 1e-9
    L = rng.uniform(0.4, 1.3)
    t_hit = rng.choice([rng.uniform(0.0, 1.0), rng.uniform(0.9, 1.1), rng.uniform(-0.1, 0.1)])
    frm = surf - dirn * L * t_hit + rng.normal(size=3) * rng.choice([0.0, 0.002])
    to = frm + dirn * L
    return frm.astype(np.float32), to.astype(np.float32), np.float32(BALL_R if sphere else 0.0), pos, R.reshape(9).copy()


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    stats = {"both": 0, "none": 0, "flag": 0}; worst = np.zeros(2); bad = []
    for i in range(n):
        frm, to, rad, pos, R = case(rng)
        a = np.zeros(4, np.float32); b = np.zeros(4, np.float32)
        ha = ref.ref_ray_convex(p(frm), p(to), p(HALF), rad, p(pos), p(R), p(a))
        hb = port.port_ray_convex(p(frm), p(to), p(HALF), rad, p(pos), p(R), p(b))
        if ha != hb:
            stats["flag"] += 1; bad.append((i, "flag", ha, hb, a, b)); continue
        if not ha:
            stats["none"] += 1; continue
        stats["both"] += 1
        e = np.array([abs(a[0] - b[0]), np.abs(a[1:] - b[1:]).max()])
        worst = np.maximum(worst, e)
        if e.max() > 0: bad.append((i, "value", *e, float(rad)))
    print(stats, "worst |dfrac| %.3g |dn| %.3g" % tuple(worst))
    for x in bad[:20]: print("  ", x)
    print(len(bad), "cases not bit-identical")


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()
