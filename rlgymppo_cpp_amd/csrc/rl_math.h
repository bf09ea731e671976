// rl_math.h — small fp32 vector/matrix/quaternion toolkit shared by the host and gfx950 builds of the
// arena stepper.  Every function is `RLG_HD` (host+device under hipcc, plain inline under g++), uses
// only fp32, and spells out the operation order (no reliance on contraction: build with
// -ffp-contract=off) so the host build and the device build agree to rounding of the libm calls.
//
// Conventions follow the reference's Bullet types so the restated algorithms read the same:
//   M3 is ROW-major like btMatrix3x3 (LinearMath/btMatrix3x3.h); the car basis COLUMNS are
//   forward/right/up (RocketSim MathTypes.h:162).
#pragma once
// static branch hints: the tick's code is far bigger than the instruction cache and every wavefront runs alone on its SIMD, so a taken branch
// into a cold cache line is a stall nothing hides -- the rare sides of the tick's branches are marked so that the common path is laid out straight
#define RLG_LIKELY(x) __builtin_expect(!!(x), 1)
#define RLG_UNLIKELY(x) __builtin_expect(!!(x), 0)
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RLG_HD __host__ __device__ __forceinline__
// Big routines are real calls on the device: fully inlined, one gym step is ~330 KB of code and every wavefront streams
// it through the 64 KB instruction cache on every tick.
#define RLG_HD_NOINLINE __host__ __device__ __noinline__ inline  /* `inline` only for ODR linkage of header definitions */
// rarely executed routines (overflow fallbacks, respawns, car-car bumps): real calls, and `cold` so that the branches into them are laid out off the common path
#define RLG_HD_COLD __host__ __device__ __noinline__ __attribute__((cold)) inline
// The tick's small per-phase routines (a few hundred instructions, called once per tick) are inlined: as real calls they cost 2 % of a
// collection launch and 11 % of its scratch write-back in register saves (-DRLG_NO_INLINE_SMALL restores the calls).  Inlining the mid-size
// ones (wheel rays, solver_prepare: 1.3 - 2.8 K instructions) takes another 19 % off the write-back at equal time (-DRLG_NO_INLINE_MID);
// the big per-phase ones (car_pre_tick_finish, collide_body, collide_merge: 2 - 6 K instructions, one call site each) another 55 % and
// 1.7 % of the launch (-DRLG_NO_INLINE_BIG).  What stays a call: routines with several call sites or rare execution (GJK, EPA, box-box,
// the order emulation) -- inlining ALL of the penetration-depth code once cost 50 % of the launch in instruction-cache misses.
#ifdef RLG_NO_INLINE_SMALL
#define RLG_HD_SMALL RLG_HD_NOINLINE
#else
#define RLG_HD_SMALL RLG_HD
#endif
#ifdef RLG_NO_INLINE_MID
#define RLG_HD_MID RLG_HD_NOINLINE
#else
#define RLG_HD_MID RLG_HD
#endif
#ifdef RLG_NO_INLINE_BIG
#define RLG_HD_BIG RLG_HD_NOINLINE
#else
#define RLG_HD_BIG RLG_HD
#endif
#ifdef RLG_NO_INLINE_T4
#define RLG_HD_T4 RLG_HD_NOINLINE
#else
#define RLG_HD_T4 RLG_HD
#endif
#ifdef RLG_NO_INLINE_T5
#define RLG_HD_T5 RLG_HD_NOINLINE
#else
#define RLG_HD_T5 RLG_HD
#endif
#ifdef RLG_NO_INLINE_T7
#define RLG_HD_T7 RLG_HD_NOINLINE
#else
#define RLG_HD_T7 RLG_HD
#endif
#ifdef RLG_INLINE_T9
#define RLG_HD_T9 RLG_HD
#else
#define RLG_HD_T9 RLG_HD_NOINLINE
#endif
#ifdef RLG_INLINE_T6A   /* take_snapshot / event_tracker_update stay calls: with them AND the T6B pair inlined the 2v2 collection kernel faults on the GPU (either pair alone is fine and T6B carries the gain) */
#define RLG_HD_T6A RLG_HD
#else
#define RLG_HD_T6A RLG_HD_NOINLINE
#endif
#ifdef RLG_NO_INLINE_T6B
#define RLG_HD_T6B RLG_HD_NOINLINE
#else
#define RLG_HD_T6B RLG_HD
#endif
#define RLG_NOUNROLL _Pragma("nounroll")
#define RLG_UNROLL _Pragma("unroll")
// The stepper kernels keep each env's state and tick scratch in LDS.  Out-of-line device functions receive them through
// generic pointers (flat_load/flat_store, no alias information); this assumption lets LLVM's InferAddressSpaces turn
// those accesses into ds_read/ds_write.  Only valid where EVERY device caller passes an LDS object.
#if defined(__HIP_DEVICE_COMPILE__)
#define RLG_ASSUME_LDS(ref) __builtin_assume(__builtin_amdgcn_is_shared((const void*)&(ref)))
#define RLG_ASSUME_LDS_W(ref) do { if constexpr (BIG == 0) RLG_ASSUME_LDS(ref); } while (0)   // (a TickWork<NC, BIG>: the big one lives in global memory)
#else
#define RLG_ASSUME_LDS(ref) ((void)0)
#define RLG_ASSUME_LDS_W(ref) ((void)0)
#endif
#else
#define RLG_NOUNROLL
#define RLG_UNROLL
#define RLG_ASSUME_LDS(ref) ((void)0)
#define RLG_ASSUME_LDS_W(ref) ((void)0)
#define RLG_HD inline
#define RLG_HD_NOINLINE inline
#define RLG_HD_COLD inline
// The tick's small per-phase routines (a few hundred instructions, called once per tick) are inlined: as real calls they cost 2 % of a
// collection launch and 11 % of its scratch write-back in register saves (-DRLG_NO_INLINE_SMALL restores the calls).  Inlining the mid-size
// ones (wheel rays, solver_prepare: 1.3 - 2.8 K instructions) takes another 19 % off the write-back at equal time (-DRLG_NO_INLINE_MID);
// the big per-phase ones (car_pre_tick_finish, collide_body, collide_merge: 2 - 6 K instructions, one call site each) another 55 % and
// 1.7 % of the launch (-DRLG_NO_INLINE_BIG).  What stays a call: routines with several call sites or rare execution (GJK, EPA, box-box,
// the order emulation) -- inlining ALL of the penetration-depth code once cost 50 % of the launch in instruction-cache misses.
#ifdef RLG_NO_INLINE_SMALL
#define RLG_HD_SMALL RLG_HD_NOINLINE
#else
#define RLG_HD_SMALL RLG_HD
#endif
#ifdef RLG_NO_INLINE_MID
#define RLG_HD_MID RLG_HD_NOINLINE
#else
#define RLG_HD_MID RLG_HD
#endif
#ifdef RLG_NO_INLINE_BIG
#define RLG_HD_BIG RLG_HD_NOINLINE
#else
#define RLG_HD_BIG RLG_HD
#endif
#ifdef RLG_NO_INLINE_T4
#define RLG_HD_T4 RLG_HD_NOINLINE
#else
#define RLG_HD_T4 RLG_HD
#endif
#ifdef RLG_NO_INLINE_T5
#define RLG_HD_T5 RLG_HD_NOINLINE
#else
#define RLG_HD_T5 RLG_HD
#endif
#ifdef RLG_NO_INLINE_T7
#define RLG_HD_T7 RLG_HD_NOINLINE
#else
#define RLG_HD_T7 RLG_HD
#endif
#ifdef RLG_INLINE_T9
#define RLG_HD_T9 RLG_HD
#else
#define RLG_HD_T9 RLG_HD_NOINLINE
#endif
#ifdef RLG_INLINE_T6A   /* take_snapshot / event_tracker_update stay calls: with them AND the T6B pair inlined the 2v2 collection kernel faults on the GPU (either pair alone is fine and T6B carries the gain) */
#define RLG_HD_T6A RLG_HD
#else
#define RLG_HD_T6A RLG_HD_NOINLINE
#endif
#ifdef RLG_NO_INLINE_T6B
#define RLG_HD_T6B RLG_HD_NOINLINE
#else
#define RLG_HD_T6B RLG_HD
#endif
#endif
#include "rl_libm.h"   // rl_sinf / rl_cosf / rl_atan2f / rl_asinf: the same bits on the device and on the host (= glibc's)
// Phase stamps for the tick profiler build (tools/prof_cycles.py); empty in product builds.
#ifndef RLG_PROF
#define RLG_PROF(i) ((void)0)
#endif
#ifndef RLG_SPROF   // sub-phase stamps of the -DRLG_FINE_PROF build (tools/fine_prof.py buckets 32..63)
#define RLG_SPROF(i) ((void)0)
#endif

namespace rlg {

constexpr float SIMD_EPS = 1.1920928955078125e-7f;  // FLT_EPSILON (btScalar.h SIMD_EPSILON)
constexpr float PI_F = 3.14159265358979323846f;

struct V3 {
    float x, y, z;
};

RLG_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RLG_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
RLG_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
RLG_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
RLG_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
RLG_HD V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
RLG_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
// Two divisions by a scalar, because the reference has two vector types: RocketSim's Vec divides every component (MathTypes.cpp:11-22),
// btVector3 multiplies by the reciprocal (btVector3.h:210-226, 852-865; safeNormalize too).  One ulp apart; every site says which one it
// restates.  btVector3::normalize itself cannot be restated exactly: the reference's x86 build takes the SSE branch (btScalar.h:216-223,
// btVector3.h:308-346: rsqrtss + one Newton step, a hardware-specific approximation), `normalized` below is the portable branch.
RLG_HD V3 vdiv_rs(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
RLG_HD V3 vdiv_bt(V3 a, float s) { const float r = 1.0f / s; return v3(a.x * r, a.y * r, a.z * r); }
RLG_HD V3& operator+=(V3& a, V3 b) { a = a + b; return a; }
RLG_HD V3& operator-=(V3& a, V3 b) { a = a - b; return a; }
RLG_HD V3& operator*=(V3& a, float s) { a = a * s; return a; }
RLG_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RLG_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RLG_HD float len2(V3 a) { return dot(a, a); }
RLG_HD float len(V3 a) { return sqrtf(dot(a, a)); }
RLG_HD bool is_zero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
// btVector3::normalize / normalized as the reference's x86 build computes it (btVector3.h:308-346, the SSE branch): the reciprocal
// square root is `rsqrtss` + one Newton step, not a division by the length.  The instruction's result is hardware-defined; on the Intel
// core that recorded the fixtures it is 1 / sqrt(midpoint of the argument's 2^-10-wide mantissa bucket) rounded to 12 bits, which the
// routine below reproduces for every normal argument (checked against the instruction on 2e8 inputs) -- on the host and on the device.
// Effect: a unit vector normalises to 0.99999994 of itself, as in the reference (the floor normal under a resting car).
RLG_HD float rsqrtss_emulated(float x) {
    const float mid = rl_u2f((rl_f2u(x) & 0xFFFFE000u) | 0x1000u);
    const float r = 1.0f / sqrtf(mid);
    return rl_u2f((rl_f2u(r) + 0x400u) & 0xFFFFF800u);
}
RLG_HD V3 normalized(V3 a) {
    float vd = a.x * a.x; vd = vd + a.y * a.y; vd = vd + a.z * a.z;
    float y = rsqrtss_emulated(vd);
    vd = vd * 0.5f; vd = vd * y; vd = vd * y;
    y = y * (1.5f - vd);
    return v3(a.x * y, a.y * y, a.z * y);
}
// btVector3::safeNormalize (btVector3.h:287-300): (1,0,0) when shorter than eps
RLG_HD V3 safe_normalized(V3 a) {
    float l2 = len2(a);
    if (l2 >= SIMD_EPS * SIMD_EPS) return vdiv_bt(a, sqrtf(l2));
    return v3(1.f, 0.f, 0.f);
}
RLG_HD float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }
RLG_HD float sgnf(float v) { return (float)((v > 0.f) - (v < 0.f)); }
RLG_HD float get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

struct M3 {
    V3 r0, r1, r2;  // rows
};
RLG_HD M3 m3_rows(V3 a, V3 b, V3 c) { M3 m; m.r0 = a; m.r1 = b; m.r2 = c; return m; }
RLG_HD M3 m3_cols(V3 c0, V3 c1, V3 c2) { return m3_rows(v3(c0.x, c1.x, c2.x), v3(c0.y, c1.y, c2.y), v3(c0.z, c1.z, c2.z)); }
RLG_HD M3 m3_identity() { return m3_rows(v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)); }
RLG_HD V3 col0(const M3& m) { return v3(m.r0.x, m.r1.x, m.r2.x); }
RLG_HD V3 col1(const M3& m) { return v3(m.r0.y, m.r1.y, m.r2.y); }
RLG_HD V3 col2(const M3& m) { return v3(m.r0.z, m.r1.z, m.r2.z); }
RLG_HD V3 operator*(const M3& m, V3 v) { return v3(dot(m.r0, v), dot(m.r1, v), dot(m.r2, v)); }
// v * M  (btVector3 * btMatrix3x3 == M^T v)
RLG_HD V3 tmul(const M3& m, V3 v) { return v3(dot(col0(m), v), dot(col1(m), v), dot(col2(m), v)); }
RLG_HD M3 transpose(const M3& m) { return m3_rows(col0(m), col1(m), col2(m)); }
RLG_HD M3 operator*(const M3& a, const M3& b) {
    V3 c0 = col0(b), c1 = col1(b), c2 = col2(b);
    return m3_rows(v3(dot(a.r0, c0), dot(a.r0, c1), dot(a.r0, c2)), v3(dot(a.r1, c0), dot(a.r1, c1), dot(a.r1, c2)),
                   v3(dot(a.r2, c0), dot(a.r2, c1), dot(a.r2, c2)));
}
// btMatrix3x3::inverse (btMatrix3x3.h:1093-1103): cofactors over the determinant
RLG_HD M3 m3_inverse(const M3& m) {
    const float e[3][3] = {{m.r0.x, m.r0.y, m.r0.z}, {m.r1.x, m.r1.y, m.r1.z}, {m.r2.x, m.r2.y, m.r2.z}};
    auto cofac = [&](int r1, int c1, int r2, int c2) { return e[r1][c1] * e[r2][c2] - e[r1][c2] * e[r2][c1]; };
    const V3 co = v3(cofac(1, 1, 2, 2), cofac(1, 2, 2, 0), cofac(1, 0, 2, 1));
    const float det = dot(m.r0, co);
    const float s = 1.0f / det;
    return m3_rows(v3(co.x * s, cofac(0, 2, 2, 1) * s, cofac(0, 1, 1, 2) * s),
                   v3(co.y * s, cofac(0, 0, 2, 2) * s, cofac(0, 2, 1, 0) * s),
                   v3(co.z * s, cofac(0, 1, 2, 0) * s, cofac(0, 0, 1, 1) * s));
}
// btMatrix3x3::scaled(s): column i scaled by s[i]
RLG_HD M3 scaled_cols(const M3& m, V3 s) { return m3_rows(m.r0 * s, m.r1 * s, m.r2 * s); }

struct Q4 {
    float x, y, z, w;
};
// The quaternion / matrix conversions and products below follow the SSE branches of Bullet's headers, which is what the reference's
// x86 build runs (btScalar.h:216-223): same products as the portable branches, but summed in another order, so the last bit differs.
// Each is checked bit for bit against the reference's own inline functions (tests/cpp/bullet_math_check.cpp).
// btQuaternion::dot / length2 (btQuaternion.h:336-353): (x x + z z) + (y y + w w)
RLG_HD float qlen2(Q4 q) { return (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w); }
// btMatrix3x3::getRotation (btMatrix3x3.h:421-487): the four numerators first, all scaled by 0.5 / sqrt(x) at the end
RLG_HD Q4 m3_to_quat(const M3& m) {
    float trace = m.r0.x + m.r1.y + m.r2.z;
    float t[4]; float x;
    if (trace > 0.f) {
        x = trace + 1.0f;
        t[0] = m.r2.y - m.r1.z; t[1] = m.r0.z - m.r2.x; t[2] = m.r1.x - m.r0.y; t[3] = x;
    } else {
        // (i = the largest diagonal entry; the three cases written out -- `rows[i]` and `t[j]` behind run-time indices put both arrays into scratch memory
        // on the device, and a car that faces backwards takes this branch on every tick)
        const int i = m.r0.x < m.r1.y ? (m.r1.y < m.r2.z ? 2 : 1) : (m.r0.x < m.r2.z ? 2 : 0);
        if (i == 0) {        // j = 1, k = 2
            x = m.r0.x - m.r1.y - m.r2.z + 1.0f;
            t[3] = m.r2.y - m.r1.z; t[1] = m.r1.x + m.r0.y; t[2] = m.r2.x + m.r0.z; t[0] = x;
        } else if (i == 1) { // j = 2, k = 0
            x = m.r1.y - m.r2.z - m.r0.x + 1.0f;
            t[3] = m.r0.z - m.r2.x; t[2] = m.r2.y + m.r1.z; t[0] = m.r0.y + m.r1.x; t[1] = x;
        } else {             // j = 0, k = 1
            x = m.r2.z - m.r0.x - m.r1.y + 1.0f;
            t[3] = m.r1.x - m.r0.y; t[0] = m.r0.z + m.r2.x; t[1] = m.r1.z + m.r2.y; t[2] = x;
        }
    }
    const float s = 0.5f / sqrtf(x);
    Q4 q; q.x = t[0] * s; q.y = t[1] * s; q.z = t[2] * s; q.w = t[3] * s;
    return q;
}
// btMatrix3x3::setRotation (btMatrix3x3.h:216-272): raw products, summed, times 2 / |q|^2, plus the identity
RLG_HD M3 quat_to_m3(Q4 q) {
    const float s = 2.0f / qlen2(q);
    const float X = q.x, Y = q.y, Z = q.z, W = q.w;
    return m3_rows(v3((-(Y * Y) - Z * Z) * s + 1.0f, (X * Y - W * Z) * s + 0.0f, (Z * X + Y * W) * s + 0.0f),
                   v3((X * Y + Z * W) * s + 0.0f, (-(X * X) - Z * Z) * s + 1.0f, (Y * Z - W * X) * s + 0.0f),
                   v3((Z * X - W * Y) * s + 0.0f, (Y * Z + W * X) * s + 0.0f, (-(X * X) - Y * Y) * s + 1.0f));
}
// btQuaternion operator*(q1, q2) (btQuaternion.h:619-650): (w1 v2 - v1 x' v2) + (v1 w2 + v1 x'' v2) component by component
RLG_HD Q4 qmul(Q4 a, Q4 b) {
    Q4 r;
    r.x = (a.w * b.x - a.z * b.y) + (a.x * b.w + a.y * b.z);
    r.y = (a.w * b.y - a.x * b.z) + (a.y * b.w + a.z * b.x);
    r.z = (a.w * b.z - a.y * b.x) + (a.z * b.w + a.x * b.y);
    r.w = (a.w * b.w - a.z * b.z) - (a.x * b.x + a.y * b.y);
    return r;
}
// rotation about a unit axis (btQuaternion(axis, angle))
RLG_HD Q4 quat_axis_angle(V3 axis, float angle) {
    float d = len(axis);
    float s = rl_sinf(angle * 0.5f) / d;
    Q4 q; q.x = axis.x * s; q.y = axis.y * s; q.z = axis.z * s; q.w = rl_cosf(angle * 0.5f);
    return q;
}

// quatRotate (btQuaternion.h): q * v * q^-1 through the quaternion products Bullet uses
RLG_HD V3 quat_rotate(Q4 r, V3 v) {
    // q = rotation * v  (btQuaternion operator*(q, w))
    Q4 q;
    q.x = r.w * v.x + r.y * v.z - r.z * v.y;
    q.y = r.w * v.y + r.z * v.x - r.x * v.z;
    q.z = r.w * v.z + r.x * v.y - r.y * v.x;
    q.w = -r.x * v.x - r.y * v.y - r.z * v.z;
    Q4 inv; inv.x = -r.x; inv.y = -r.y; inv.z = -r.z; inv.w = r.w;   // rotation.inverse()
    Q4 o = qmul(q, inv);
    return v3(o.x, o.y, o.z);
}

// btTransformUtil::integrateTransform's rotation part (LinearMath/btTransformUtil.h:37-87)
RLG_HD_T7 M3 integrate_rotation(const M3& basis, V3 angvel, float dt) {
    const float ANGULAR_MOTION_THRESHOLD = 0.5f * (PI_F * 0.5f);
    float fAngle2 = len2(angvel);
    float fAngle = 0.f;
    if (fAngle2 > SIMD_EPS) fAngle = sqrtf(fAngle2);
    if (fAngle * dt > ANGULAR_MOTION_THRESHOLD) fAngle = ANGULAR_MOTION_THRESHOLD / dt;
    V3 axis;
    if (fAngle < 0.001f)
        axis = angvel * (0.5f * dt - (dt * dt * dt) * 0.020833333333f * fAngle * fAngle);
    else
        axis = angvel * (rl_sinf(0.5f * fAngle * dt) / fAngle);
    Q4 dorn; dorn.x = axis.x; dorn.y = axis.y; dorn.z = axis.z; dorn.w = rl_cosf(fAngle * dt * 0.5f);
    Q4 orn0 = m3_to_quat(basis);
    Q4 p = qmul(dorn, orn0);
    if (qlen2(p) > SIMD_EPS) {        // btQuaternion::safeNormalize -> normalize (btQuaternion.h:374-402): times 1 / sqrt(length2)
        const float inv = 1.f / sqrtf(qlen2(p));
        p.x *= inv; p.y *= inv; p.z *= inv; p.w *= inv;
    }
    if (qlen2(p) > SIMD_EPS) return quat_to_m3(p);
    return basis;
}

// btPlaneSpace1 (btVector3.h)
RLG_HD void plane_space1(V3 n, V3& p, V3& q) {
    const float SIMDSQRT12 = 0.7071067811865475244008443621048490f;
    if (fabsf(n.z) > SIMDSQRT12) {
        float a = n.y * n.y + n.z * n.z;
        float k = 1.f / sqrtf(a);
        p = v3(0.f, -n.z * k, n.y * k);
        q = v3(a * k, -n.x * p.z, n.x * p.y);
    } else {
        float a = n.x * n.x + n.y * n.y;
        float k = 1.f / sqrtf(a);
        p = v3(-n.y * k, n.x * k, 0.f);
        q = v3(-n.z * p.y, n.z * p.x, a * k);
    }
}

// ---- counter-based RNG (Philox4x32-10) : one stream per env, no carried generator state --------
struct Philox {
    uint32_t key0, key1;
    uint32_t c0, c1, c2, c3;
};
RLG_HD void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    const uint64_t M0 = 0xD2511F53ull, M1 = 0xCD9E8D57ull;
    uint64_t p0 = M0 * (uint64_t)c0, p1 = M1 * (uint64_t)c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
// 4 x u32 for (seed, stream, counter)
RLG_HD void philox4(uint32_t seed_lo, uint32_t seed_hi, uint32_t stream, uint32_t ctr_lo, uint32_t ctr_hi, uint32_t out[4]) {
    uint32_t c0 = ctr_lo, c1 = ctr_hi, c2 = stream, c3 = 0x9E3779B9u;
    uint32_t k0 = seed_lo, k1 = seed_hi;
    for (int i = 0; i < 10; i++) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// ---- the reference's own engine (parity tests: RlgpuArenaHidden::ref_engine) ---------------------------------------------------------------
// std::default_random_engine of libstdc++ = minstd_rand0: x <- 16807 x mod (2^31 - 1), operator() returns the new state, min() = 1, max() = 2^31 - 2.
// The formulas are RocketSim's (Math.cpp:44-57) and libstdc++'s (bits/uniform_int_dist.h, bits/stl_algo.h), restated; pinned draw for draw against the
// real reference with its thread engine assigned a known state (tests/golden/make_rng_golden.py).
struct RefEngine {
    uint32_t x;
    RLG_HD uint32_t next() {
        const uint64_t p = (uint64_t)x * 16807ull;                     // < 2^46
        uint32_t r = (uint32_t)(p & 0x7fffffffull) + (uint32_t)(p >> 31);   // 2^31 = 1 (mod 2^31 - 1)
        if (r >= 0x7fffffffu) r -= 0x7fffffffu;
        x = r; return r;
    }
    // Math::RandFloat(min, max) = min + (engine() / (float)engine.max()) * (max - min)   (engine.max() = 2147483646 is 2^31 as a float)
    RLG_HD float uni(float lo, float hi) { const float u = (float)next() / 2147483648.f; return lo + u * (hi - lo); }
    // Math::RandInt(min, max) = min + engine() % (max - min)
    RLG_HD int rand_int(int lo, int hi) { return lo + (int)(next() % (uint32_t)(hi - lo)); }
    // std::uniform_int_distribution<unsigned long>(0, hi)(engine): the engine's range 2^31 - 3 is scaled down and draws past the last whole bucket are redrawn
    RLG_HD uint32_t uniform_int(uint32_t hi) {
        const uint32_t ue = hi + 1u, scaling = 2147483645u / ue, past = ue * scaling;
        uint32_t r;
        do r = next() - 1u; while (r >= past);
        return r / scaling;
    }
    // std::shuffle(a, a + 5, engine): an odd count goes in pairs, one draw for two swap positions (__gen_two_uniform_ints)
    RLG_HD void shuffle5(int (&a)[5]) {
        for (int i = 1; i < 5; i += 2) {
            const uint32_t b0 = (uint32_t)i + 1u, b1 = b0 + 1u, xx = uniform_int(b0 * b1 - 1u);
            const int p0 = (int)(xx / b1), p1 = (int)(xx % b1);
            int t = a[i]; a[i] = a[p0]; a[p0] = t;
            t = a[i + 1]; a[i + 1] = a[p1]; a[p1] = t;
        }
    }
};

RLG_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }   // bit pattern (monotonic for f >= 0)
RLG_HD float u32_to_unit(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }  // [0,1)

}  // namespace rlg
