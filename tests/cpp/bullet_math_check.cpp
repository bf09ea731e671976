// bullet_math_check.cpp — csrc/rl_math.h against the reference's own inline math (Bullet's LinearMath headers under
// /root/reference, compiled as the reference compiles them: SSE branches on x86), bit for bit, on random arguments:
//   btVector3::normalize (rsqrtss + one Newton step)  vs  normalized()        [exact on the Intel core that recorded the fixtures]
//   btQuaternion(axis, angle), q1 * q2, q * v, quatRotate, btMatrix3x3(q) = setRotation, getRotation, btMatrix3x3(q) * v, safeNormalize
// Build container only (needs /root/reference); tests/test_host_cpp.py runs it when the reference is mounted.
#include <cstdio>
#include <cstring>
#include <cmath>
#include <bullet3-3.24/LinearMath/btQuaternion.h>
#include <bullet3-3.24/LinearMath/btMatrix3x3.h>
#undef SIMDSQRT12
#include "../../rlgymppo_cpp_amd/csrc/rl_math.h"
using namespace rlg;
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static bool same(float a, float b) { return bits(a) == bits(b) || (a == 0.f && b == 0.f); }
int main() {
    unsigned seed = 1; auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.f * 2.f - 1.f; };
    long bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int N = 300000;
    for (int it = 0; it < N; it++) {
        V3 e = v3(rnd() * 30, rnd() * 30, rnd() * 30); float ang = rnd() * 3.f; V3 v = v3(rnd(), rnd(), rnd());
        btVector3 be(e.x, e.y, e.z), bv(v.x, v.y, v.z);
        btVector3 bn = bv; bn.normalize(); V3 n = normalized(v);
        if (!same(bn.x(), n.x) || !same(bn.y(), n.y) || !same(bn.z(), n.z)) bad[0]++;
        btQuaternion bq(be, ang); Q4 q = quat_axis_angle(e, ang);
        if (!same(bq.x(), q.x) || !same(bq.y(), q.y) || !same(bq.z(), q.z) || !same(bq.w(), q.w)) bad[1]++;
        btQuaternion bq2(btVector3(rnd(), rnd(), rnd()), rnd() * 3.f); Q4 q2; q2.x = bq2.x(); q2.y = bq2.y(); q2.z = bq2.z(); q2.w = bq2.w();
        btQuaternion bp = bq * bq2; Q4 p = qmul(q, q2);
        if (!same(bp.x(), p.x) || !same(bp.y(), p.y) || !same(bp.z(), p.z) || !same(bp.w(), p.w)) bad[2]++;
        btVector3 bqr = quatRotate(bq, bv); V3 qr = quat_rotate(q, v);
        if (!same(bqr.x(), qr.x) || !same(bqr.y(), qr.y) || !same(bqr.z(), qr.z)) bad[3]++;
        btMatrix3x3 bm(bq); M3 m = quat_to_m3(q);
        bool ok = true;
        for (int r = 0; r < 3; r++) { const V3 row = r == 0 ? m.r0 : r == 1 ? m.r1 : m.r2; if (!same(bm[r].x(), row.x) || !same(bm[r].y(), row.y) || !same(bm[r].z(), row.z)) ok = false; }
        if (!ok) bad[4]++;
        btVector3 br = bm * bv; V3 r = m * v;
        if (!same(br.x(), r.x) || !same(br.y(), r.y) || !same(br.z(), r.z)) bad[5]++;
        btQuaternion bg; bm.getRotation(bg); Q4 g = m3_to_quat(m);
        if (!same(bg.x(), g.x) || !same(bg.y(), g.y) || !same(bg.z(), g.z) || !same(bg.w(), g.w)) bad[6]++;
        btQuaternion bs = bp; bs.safeNormalize(); Q4 sn = p; if (qlen2(sn) > SIMD_EPS) { const float inv = 1.f / sqrtf(qlen2(sn)); sn.x *= inv; sn.y *= inv; sn.z *= inv; sn.w *= inv; }
        if (!same(bs.x(), sn.x) || !same(bs.y(), sn.y) || !same(bs.z(), sn.z) || !same(bs.w(), sn.w)) bad[7]++;
    }
    printf("%d cases: mismatches normalize %ld, quaternion(axis, angle) %ld, q*q %ld, quatRotate %ld, setRotation %ld, matrix*vector %ld, getRotation %ld, safeNormalize %ld\n",
           N, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5], bad[6], bad[7]);
    long t = 0; for (long b : bad) t += b;
    return t ? 1 : 0;
}
