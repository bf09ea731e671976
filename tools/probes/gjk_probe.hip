// How many cycles does ONE hitbox-triangle GJK run cost a wavefront, in isolation?  (1, 2, 8 active lanes with different poses.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../rlgymppo_cpp_amd/csrc/arena_gym.h"
using namespace rlg;
__global__ void k_probe(const MeshTri* tris, int active, int reps, unsigned long long* out, float* sink) {
    const int lane = threadIdx.x;
    float acc = 0.f; unsigned long long cyc = 0; int hits = 0;
    for (int r = 0; r < reps; r++) {
        // a hitbox-sized box tumbling just above / touching the floor triangle; every lane its own pose
        const float ang = 0.37f * (lane + 1) + 0.011f * r, tilt = 0.6f + 0.05f * lane;
        M3 R = euler_to_rot(ang, tilt, 0.3f * lane);
        V3 bc = v3(3.f + lane, -2.f, 0.55f + 0.01f * (r % 7) + 0.02f * lane);
        GjkOut g; bool deep = false;
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        bool hit = false;
        if (lane < active) hit = gjk_box_triangle(bc, R, hitbox_core(), BOX_MARGIN, tris[0], CBT_CAR, g, deep);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        cyc += t1 - t0;
        if (hit) { acc += g.dist + g.n.z; hits++; }
    }
    if (lane == 0) { out[0] = cyc / reps; }
    if (lane < active) { out[1 + lane] = hits; sink[lane] = acc; }
}
int main() {
    MeshTri t{}; 
    t.v0x = -80; t.v0y = -100; t.v0z = 0; t.v1x = 80; t.v1y = -100; t.v1z = 0; t.v2x = -80; t.v2y = 100; t.v2z = 0;   // a floor triangle, counter-clockwise from above
    MeshTri* d; hipMalloc(&d, sizeof(t)); hipMemcpy(d, &t, sizeof(t), hipMemcpyHostToDevice);
    unsigned long long* out; hipMalloc(&out, 72 * 8); float* sink; hipMalloc(&sink, 64 * 4);
    for (int active : {1, 2, 4, 8, 16}) {
        hipMemset(out, 0, 72 * 8);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d, active, 200, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[72]; hipMemcpy(h, out, 72 * 8, hipMemcpyDeviceToHost);
        std::printf("%2d active lanes: %llu cycles per call of the wavefront; contacts of lane 0: %llu / 200\n", active, h[0], h[1]);
    }
    return 0;
}
