// infer_probe.hip -- the in-kernel policy inference of the collection kernel (csrc/infer_device.h:wave_infer) alone: every wavefront infers 8 rows
// `iters` times; cycles per inference from s_memtime, wall time from events.  Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I rlgymppo_cpp_amd/csrc
//   [-DRLINFER_BUFS=2|3] [-DPROBE_NO_HEAD] tools/probes/infer_probe.hip -o tools/probes/infer_probe ;  run: infer_probe [waves] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "infer_device.h"
using namespace rlinfer;
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_inf(InferNet net, HeadArgs head, const float* obs, int n_rows, int iters, unsigned long long* cyc) {
    __shared__ short buf0[8 * 264 * 2], buf1[8 * 264 * 2];
    __shared__ unsigned char pad[36 * 1024];   // (as much LDS as a collection workgroup holds: four workgroups per CU)
    int picked[8];
    if (threadIdx.x == 9999) pad[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        HeadArgs h = head; h.call_ctr += it;
        h.actions = head.actions + (size_t)blockIdx.x * 8; h.logp = head.logp + (size_t)blockIdx.x * 8;
        wave_infer<8>(net, h, obs + (size_t)blockIdx.x * 8 * net.D, 0, n_rows, buf0, buf1, threadIdx.x & 63, picked);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const int waves = argc > 1 ? atoi(argv[1]) : 1024, iters = argc > 2 ? atoi(argv[2]) : 50;
    const int D = 89, dims[5] = {96, 256, 256, 256, 90};
    InferNet net{}; net.n_layers = 4; net.D = D; net.ld = 264; net.fp32 = 0;
    for (int i = 0; i < 4; i++) {
        const int K = dims[i], N = dims[i + 1], Npad = (N + 31) / 32 * 32;
        net.K[i] = K; net.N[i] = N; net.Npad[i] = Npad;
        std::vector<unsigned short> w((size_t)Npad * K);
        for (auto& x : w) { float f = ((rand() % 2001) - 1000) * 1e-4f; unsigned u; memcpy(&u, &f, 4); x = (unsigned short)(u >> 16); }
        std::vector<float> b(Npad); for (auto& x : b) x = ((rand() % 201) - 100) * 1e-3f;
        short* dw; float* db;
        CK(hipMalloc(&dw, w.size() * 2)); CK(hipMemcpy(dw, w.data(), w.size() * 2, hipMemcpyHostToDevice));
        CK(hipMalloc(&db, b.size() * 4)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
        net.W[i] = dw; net.bias[i] = db;
    }
    std::vector<float> obs((size_t)waves * 8 * D); for (auto& x : obs) x = ((rand() % 2001) - 1000) * 1e-3f;
    float* dobs; CK(hipMalloc(&dobs, obs.size() * 4)); CK(hipMemcpy(dobs, obs.data(), obs.size() * 4, hipMemcpyHostToDevice));
    HeadArgs head{}; head.A = 90; head.inv_temp = 1.f; head.deterministic = 0; head.noise = nullptr; head.seed_lo = 1; head.seed_hi = 2; head.call_ctr = 0; head.probs_out = nullptr;
    CK(hipMalloc(&head.actions, (size_t)waves * 8 * 4)); CK(hipMalloc(&head.logp, (size_t)waves * 8 * 4));
    unsigned long long* dcyc; CK(hipMalloc(&dcyc, (size_t)waves * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_inf, dim3(waves), dim3(64), 0, 0, net, head, (const float*)dobs, 8, iters, dcyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(waves); CK(hipMemcpy(cyc.data(), dcyc, (size_t)waves * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (auto c : cyc) mean += (double)c; mean /= waves;
        printf("%d waves x %d inferences: %.3f ms, mean %.1f K cycles per inference\n", waves, iters, ms, mean / iters / 1e3);
    }
    return 0;
}
