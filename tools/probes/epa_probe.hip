// What does ONE penetration-depth query (second GJK + EPA, csrc/arena_epa.h) cost a wavefront in isolation, next to a plain GJK run?
// (one lane active; the arena in LDS as in the step kernels.)   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 epa_probe.hip -o epa_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__shared__ unsigned char* g_epa_small_ptr[1];
__shared__ unsigned char* g_epa_big_ptr[1];
__device__ int g_cnt[4];
__device__ unsigned long long g_phase[8]; __device__ unsigned long long g_last;
#define RLG_EPA_PROF(i) do { unsigned long long t_ = __builtin_amdgcn_s_memtime(); if ((i) > 0) atomicAdd(&g_phase[i], t_ - g_last); g_last = t_; } while (0)
#ifndef PROBE_V
#define PROBE_V 14
#define PROBE_F 34
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define RLG_EPA_ARENA_DECL \
    EpaArena epa_small_ = epa_arena_at(g_epa_small_ptr[0], PROBE_V, PROBE_F); EpaArena epa_bigv_ = epa_arena_at(g_epa_big_ptr[0], EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES); \
    EpaArena* epa_big_ = g_epa_big_ptr[0] ? &epa_bigv_ : nullptr;
#define RLG_EPA_SERIALIZE_BEGIN for (unsigned long long pend_ = __ballot(1); pend_; pend_ &= pend_ - 1ull) { if ((int)(threadIdx.x & 63u) == __ffsll((unsigned long long)pend_) - 1) { atomicAdd(&g_cnt[0], 1);
#define RLG_EPA_SERIALIZE_END } }
#define RLG_EPA_BIG_PASS(rc_, CALL)
#define RLG_EPA_COUNT_BIG() atomicAdd(&g_cnt[1], 1)
#endif
#include "../../rlgymppo_cpp_amd/csrc/arena_gym.h"
using namespace rlg;
__global__ void __launch_bounds__(64) k_probe(const MeshTri* tris, float z0, int reps, unsigned long long* out, float* sink, unsigned char* big) {
    __shared__ __attribute__((aligned(16))) unsigned char arena[4096];
    const int lane = threadIdx.x;
    if (lane == 0) { g_epa_small_ptr[0] = arena; g_epa_big_ptr[0] = big; }
    __syncthreads();
    float acc = 0.f; unsigned long long cyc = 0; int hits = 0;
    for (int r = 0; r < reps; r++) {
        const float ang = 0.37f + 0.011f * r, tilt = 0.6f + 0.013f * r;
        M3 R = euler_to_rot(ang, tilt, 0.3f);
        V3 bc = v3(3.f, -2.f, z0 + 0.01f * (r % 7));
        GjkOut g; bool deep = false;
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        bool hit = false;
        if (lane < 1) hit = gjk_box_triangle(bc, R, hitbox_core(), BOX_MARGIN, tris[0], CBT_CAR, g, deep);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        cyc += t1 - t0;
        if (hit) { acc += g.dist + g.n.z; hits++; }
    }
    if (lane == 0) { out[0] = cyc / reps; out[1] = hits; sink[0] = acc; }
}
int main() {
    MeshTri t{};
    t.v0x = -80; t.v0y = -100; t.v0z = 0; t.v1x = 80; t.v1y = -100; t.v1z = 0; t.v2x = -80; t.v2y = 100; t.v2z = 0;
    MeshTri* d; hipMalloc(&d, sizeof(t)); hipMemcpy(d, &t, sizeof(t), hipMemcpyHostToDevice);
    unsigned long long* out; hipMalloc(&out, 8 * 8); float* sink; hipMalloc(&sink, 64 * 4); unsigned char* big; hipMalloc(&big, 16384);
    for (float z0 : {1.0f, 0.62f, 0.45f, 0.30f}) {
        int zero[4] = {0, 0, 0, 0}; hipMemcpyToSymbol(HIP_SYMBOL(g_cnt), zero, sizeof(zero));
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, z0, 200, out, sink, big);
        hipDeviceSynchronize();
        unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost); int c[4]; hipMemcpyFromSymbol(c, HIP_SYMBOL(g_cnt), sizeof(c));
        unsigned long long ph[8]; hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph)); unsigned long long zz[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zz, sizeof(zz));
        std::printf("box centre %.2f above the floor triangle: %llu cycles per call; contacts %llu / 200; penetration-depth queries %d (full-size arena %d)\n", z0, h[0], h[1], c[0], c[1]);
        if (c[0]) std::printf("    per query: margin GJK %llu, enclose + first faces %llu, EPA loop + result %llu, witnesses %llu cycles\n", ph[1] / c[0], ph[2] / c[0], ph[3] / c[0], ph[4] / c[0]);
    }
    return 0;
}
