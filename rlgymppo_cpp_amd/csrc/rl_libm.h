// rl_libm.h — the four libm routines the physics calls, restated so that the device and the host compute the SAME bits.
//
// + - * / and sqrtf are correctly rounded on both sides (the build disables fp contraction), so the only arithmetic in which the HIP
// stepper and the host build of the same source could differ is libm: ROCm's device library and glibc round sinf / cosf / atan2f / asinf
// differently in the last bit now and then, and a contact decision that sits on the fence (the clamp of a contact normal at a
// right-angled mesh edge keeps or drops the clamped normal on the SIGN of a 1e-8 dot product) then flips between the two.  The
// reference calls glibc's routines, so these follow glibc 2.35's algorithms operation by operation -- the host functions are checked
// bit for bit against the C library over 4e8 arguments (tests/cpp/libm_check.cpp), the device functions against the host's on the GPU:
//   rl_sinf / rl_cosf   sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h (double-precision polynomials, table __sincosf_table) as
//                       the FMA build evaluates them;
//                       arguments up to 120 in magnitude (the physics stays below 7); beyond that the platform's routine answers
//   rl_atanf / rl_atan2f   flt-32/s_atanf.c, e_atan2f.c (FDLIBM, float arithmetic)
//   rl_asinf            flt-32/e_asinf.c
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef RLG_HD
#define RLG_HD inline
#endif

namespace rlg {

RLG_HD uint32_t rl_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
RLG_HD float rl_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// sinf_poly (sincosf.h): sine polynomial for even quadrants, cosine polynomial for odd ones; `flip` selects the negated coefficient set
RLG_HD float rl_sincos_poly(double x, double x2, bool flip, int n) {
    const double c0 = flip ? -0x1p0 : 0x1p0, c1 = flip ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = flip ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = flip ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = flip ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    // (every a + b * c below is ONE fused operation: x86-64 glibc dispatches to its FMA build of this code on any CPU that has the
    // instruction -- s_sinf-fma.c -- and the two builds differ next to the zeros of the functions)
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = fma(x3, s1, x);
        return (float)fma(x7, t1, s);
    }
    const double x4 = x2 * x2;
    const double t2 = fma(x2, c4, c3);
    const double t1 = fma(x2, c1, c0);
    const double x6 = x4 * x2;
    const double c = fma(x4, c2, t1);
    return (float)fma(x6, t2, c);
}
RLG_HD uint32_t rl_abstop12(float x) { return (rl_f2u(x) >> 20) & 0x7ffu; }
// reduce_fast: quadrant in bits 24..31 of the scaled product
RLG_HD double rl_reduce_fast(double x, int& n) {
    const double r = x * 0x1.45f306dc9c883p+23;
    n = ((int32_t)r + 0x800000) >> 24;
    return fma(-(double)n, 0x1.921fb54442d18p+0, x);
}
RLG_HD float rl_sinf(float y) {
    double x = y;
    if (rl_abstop12(y) < rl_abstop12(0x1.921FB6p-1f)) {
        if (rl_abstop12(y) < rl_abstop12(0x1p-12f)) return y;
        return rl_sincos_poly(x, x * x, false, 0);
    }
    if (rl_abstop12(y) < rl_abstop12(120.0f)) {
        int n; x = rl_reduce_fast(x, n);
        const double s = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;
        return rl_sincos_poly(x * s, x * x, (n & 2) != 0, n);
    }
    return sinf(y);
}
RLG_HD float rl_cosf(float y) {
    double x = y;
    if (rl_abstop12(y) < rl_abstop12(0x1.921FB6p-1f)) {
        if (rl_abstop12(y) < rl_abstop12(0x1p-12f)) return 1.0f;
        return rl_sincos_poly(x, x * x, false, 1);
    }
    if (rl_abstop12(y) < rl_abstop12(120.0f)) {
        int n; x = rl_reduce_fast(x, n);
        const double s = ((n & 3) == 0 || (n & 3) == 3) ? 1.0 : -1.0;
        return rl_sincos_poly(x * s, x * x, (n & 2) != 0, n ^ 1);
    }
    return cosf(y);
}

RLG_HD float rl_atanf(float x) {
    const uint32_t ATANHI[4] = {0x3eed6338u, 0x3f490fdau, 0x3f7b985eu, 0x3fc90fdau};
    const uint32_t ATANLO[4] = {0x31ac3769u, 0x33222168u, 0x33140fb4u, 0x33a22168u};
    const float aT0 = rl_u2f(0x3eaaaaabu) /* (the literal 3.3333334327e-01, not the 0x3eaaaaaa of its comment) */, aT1 = rl_u2f(0xbe4ccccdu), aT2 = rl_u2f(0x3e124925u), aT3 = rl_u2f(0xbde38e38u),
                aT4 = rl_u2f(0x3dba2e6eu), aT5 = rl_u2f(0xbd9d8795u), aT6 = rl_u2f(0x3d886b35u), aT7 = rl_u2f(0xbd6ef16bu),
                aT8 = rl_u2f(0x3d4bda59u), aT9 = rl_u2f(0xbd15a221u), aT10 = rl_u2f(0x3c8569d7u);
    const int32_t hx = (int32_t)rl_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {   // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        const float r = rl_u2f(ATANHI[3]) + rl_u2f(ATANLO[3]);
        return hx > 0 ? r : -r;
    }
    if (ix < 0x3ee00000) {    // |x| < 0.4375
        if (ix < 0x31000000) return x;   // |x| < 2^-29
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    const float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    const float hi = id == 0 ? rl_u2f(ATANHI[0]) : id == 1 ? rl_u2f(ATANHI[1]) : id == 2 ? rl_u2f(ATANHI[2]) : rl_u2f(ATANHI[3]);
    const float lo = id == 0 ? rl_u2f(ATANLO[0]) : id == 1 ? rl_u2f(ATANLO[1]) : id == 2 ? rl_u2f(ATANLO[2]) : rl_u2f(ATANLO[3]);
    const float r = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -r : r;
}
RLG_HD float rl_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = rl_u2f(0x3f490fdbu), pi_o_2 = rl_u2f(0x3fc90fdbu), pi = rl_u2f(0x40490fdbu), pi_lo = rl_u2f(0xb3bbbd2eu);
    const int32_t hx = (int32_t)rl_f2u(x), hy = (int32_t)rl_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return rl_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m == 0 || m == 1) return y;
        return m == 2 ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            if (m == 2) return 3.0f * pi_o_4 + tiny;
            return -3.0f * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0f;
        if (m == 1) return -0.0f;
        if (m == 2) return pi + tiny;
        return -pi - tiny;
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = rl_atanf(fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return rl_u2f(rl_f2u(z) ^ 0x80000000u);
    if (m == 2) return pi - (z - pi_lo);
    return (z - pi_lo) - pi;
}

RLG_HD float rl_asinf(float x) {
    const float pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f, pio4_hi = 0.785398185253143310546875f;
    const float p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    const int32_t hx = (int32_t)rl_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix < 0x32000000) return x;
        const float t = x * x;
        const float w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
        return x + x * w;
    }
    float w = 1.0f - fabsf(x);
    float t = w * 0.5f;
    float p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    const float s = sqrtf(t);
    if (ix >= 0x3F79999A) {
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        w = rl_u2f(rl_f2u(s) & 0xfffff000u);
        const float c = (t - w * w) / (s + w);
        const float r = p;
        p = 2.0f * s * r - (pio2_lo - 2.0f * c);
        const float q = pio4_hi - 2.0f * w;
        t = pio4_hi - (p - q);
    }
    return hx > 0 ? t : -t;
}

// rl_powf: flt-32/e_powf.c with powf_log2_data.c / exp2f_data.c (x86-64: TOINT_INTRINSICS 0, so POWF_SCALE = 1 and exp2_inline takes the SHIFT path), as the FMA
// build evaluates it -- log2(x) by a 16-entry table and a degree-5 polynomial in double, y * log2(x), 2^that by a 32-entry table and a degree-3 polynomial.
// glibc's result is within 0.82 ulp, NOT always the correctly rounded one: the double pow rounded to float that the device used through round 5 differed from
// the reference's reward in one step of 120 live rollouts by an ulp (round 6, tools/live_gym_hip.py).  The tables are the C library's own (read out of this
// image's libm.so.6; tests/cpp/libm_check.cpp compares the function with powf over 1e8 arguments).  Handled here: x a positive normal number and a result that
// neither overflows nor is subnormal -- what CommonRewards.h feeds it; everything else goes to the platform's routine.
RLG_HD float rl_powf(float x, float y) {
    static const double LT[32] = {   // {invc, logc} x 16
        0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2,
        0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
        0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2,
        0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
        0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2,
        0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3,
        0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3,
        0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
        0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5,
        0x1.0000000000000p+0, 0x0.0p+0,
        0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4,
        0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3,
        0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3,
        0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
        0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,
        0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
    };
    static const uint64_t ET[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
    };
    const uint32_t ix = rl_f2u(x), iy = rl_f2u(y);
    const uint32_t ay = iy & 0x7fffffffu;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || ay == 0u || ay >= 0x7f800000u) return powf(x, y);      // x <= 0, subnormal, inf or nan; y 0, inf or nan
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = LT[2 * i], logc = LT[2 * i + 1], z = (double)rl_u2f(iz);
    const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2, A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
    const double r = fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    double yy = fma(A0, r, A1);
    const double p = fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = fma(A4, r, y0);
    q = fma(p, r2, q);
    yy = fma(yy, r4, q);
    const double ylogx = (double)y * yy;
    if (!(ylogx < 126.0 && ylogx > -126.0)) return powf(x, y);      // overflow, underflow and subnormal results: the C library's own tail
    // exp2_inline (sign_bias 0)
    const double SHIFT = 0x1.8p+47, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    double kd = ylogx + SHIFT;
    uint64_t ki; memcpy(&ki, &kd, 8);
    kd -= SHIFT;
    const double rr = ylogx - kd;
    uint64_t t = ET[ki % 32u];
    t += ki << (52 - 5);
    double s; memcpy(&s, &t, 8);
    const double zz = fma(C0, rr, C1);
    const double rr2 = rr * rr;
    double e = fma(C2, rr, 1.0);
    e = fma(zz, rr2, e);
    e = e * s;
    return (float)e;
}

}  // namespace rlg
