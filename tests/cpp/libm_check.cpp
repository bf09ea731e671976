// libm_check.cpp — csrc/rl_libm.h against the C library, bit for bit (host build; `-m "not gpu"`).
//   libm_check [millions of random arguments per function]
// sinf / cosf over |x| < 120 (dense near the quadrant boundaries and in the physics range |x| < 7), atan2f over both signs and 40 binades
// of ratio, atanf, asinf over [-1, 1].  Exit code 0 = identical everywhere.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#define RLG_HD inline
#include "../../rlgymppo_cpp_amd/csrc/rl_libm.h"
using namespace rlg;
static uint64_t s = 0x9E3779B97F4A7C15ull;
static inline uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
static inline float uni(float a, float b) { return a + (b - a) * (float)((rnd() >> 40) * (1.0 / 16777216.0)); }
static inline float anyfloat(float maxabs) { for (;;) { float f = rl_u2f((uint32_t)rnd()); if (std::isfinite(f) && fabsf(f) < maxabs) return f; } }
int main(int argc, char** argv) {
    const long N = (argc > 1 ? atol(argv[1]) : 20) * 1000000L;
    long bad[6] = {0, 0, 0, 0, 0, 0};
    for (long i = 0; i < N; i++) {
        float x = (i & 3) == 0 ? uni(-7.f, 7.f) : (i & 3) == 1 ? uni(-120.f, 120.f) : (i & 3) == 2 ? anyfloat(120.f) : (float)((int)(rnd() % 153) - 76) * 0.78539816f + uni(-1e-3f, 1e-3f);
        if (rl_f2u(rl_sinf(x)) != rl_f2u(sinf(x))) { if (bad[0]++ < 5) printf("sinf(%a) = %a, libm %a\n", x, rl_sinf(x), sinf(x)); }
        if (rl_f2u(rl_cosf(x)) != rl_f2u(cosf(x))) { if (bad[1]++ < 5) printf("cosf(%a) = %a, libm %a\n", x, rl_cosf(x), cosf(x)); }
        float y = (i & 1) ? anyfloat(1e30f) : uni(-2.f, 2.f), z = (i & 2) ? anyfloat(1e30f) : uni(-2.f, 2.f);
        if (rl_f2u(rl_atan2f(y, z)) != rl_f2u(atan2f(y, z))) { if (bad[2]++ < 5) printf("atan2f(%a, %a) = %a, libm %a\n", y, z, rl_atan2f(y, z), atan2f(y, z)); }
        float a = (i & 1) ? uni(-8.f, 8.f) : anyfloat(1e30f);
        if (rl_f2u(rl_atanf(a)) != rl_f2u(atanf(a))) { if (bad[3]++ < 5) printf("atanf(%a) = %a, libm %a\n", a, rl_atanf(a), atanf(a)); }
        float b = (i & 1) ? uni(-1.f, 1.f) : anyfloat(1.0001f);
        float r1 = rl_asinf(b), r2 = asinf(b);
        if (rl_f2u(r1) != rl_f2u(r2) && !(std::isnan(r1) && std::isnan(r2))) { if (bad[4]++ < 5) printf("asinf(%a) = %a, libm %a\n", b, r1, r2); }
    }
    for (long i = 0; i < N; i++) {   // powf as the rewards call it (bases in (0, 2], the exponents of CommonRewards.h and anything else in +-8) and over all positive normal bases
        float x = (i & 3) == 0 ? uni(1e-6f, 1.f) : (i & 3) == 1 ? uni(0.f, 2.f) : (i & 3) == 2 ? fabsf(anyfloat(1e30f)) : (float)(rnd() % 101) * 0.01f;
        float y = (i & 12) == 0 ? 0.5f : (i & 12) == 4 ? 0.7f : (i & 12) == 8 ? uni(-8.f, 8.f) : (float)((int)(rnd() % 33) - 16) * 0.25f;
        float r1 = rl_powf(x, y), r2 = powf(x, y);
        if (rl_f2u(r1) != rl_f2u(r2) && !(std::isnan(r1) && std::isnan(r2))) { if (bad[5]++ < 5) printf("powf(%a, %a) = %a, libm %a\n", x, y, r1, r2); }
    }
    {   // the two powf values the tick uses as constants (arena_types.h)
        volatile float a = 1.f - 0.03f, dt = 1.f / 120.f, b = 1.f - 0.35f, e = (1.f / 120.f) / (1 / 120.f);
        if (rl_f2u(powf(a, dt)) != rl_f2u(0x1.ffdebcp-1f) || rl_f2u(powf(b, e)) != rl_f2u(0x1.4cccccp-1f)) { printf("powf constants differ: %a %a\n", powf(a, dt), powf(b, e)); bad[0]++; }
    }
    printf("%ld arguments per function: mismatches sinf %ld cosf %ld atan2f %ld atanf %ld asinf %ld powf %ld\n", N, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    return (bad[0] | bad[1] | bad[2] | bad[3] | bad[4] | bad[5]) ? 1 : 0;
}
