// How long does the GPU need just to START the workgroups of a short kernel?  (Design question behind the learner GEMMs: 1024 workgroups
// x 256 threads that each live ~10 us.)  Times kernels that do almost nothing, for several grid shapes / resource footprints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int LDS_BYTES, int SPIN>
__global__ void __launch_bounds__(256) k_probe(float* out, int n) {
    __shared__ char smem[LDS_BYTES > 0 ? LDS_BYTES : 1];
    if (LDS_BYTES > 0) smem[threadIdx.x] = (char)threadIdx.x;
    float v = (float)threadIdx.x;
    for (int i = 0; i < SPIN; i++) v = v * 1.0001f + 0.5f;     // SPIN dependent FMAs ~ 4 cycles each
    if (v == -1.f && out) out[blockIdx.x] = v + (LDS_BYTES > 0 ? smem[0] : 0);
}
template <class F>
float time_us(F launch, int reps = 50) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 5; i++) launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    float* d; hipMalloc(&d, 1 << 20);
    for (int wgs : {256, 512, 1024, 2048, 4096}) {
        float t0 = time_us([&] { hipLaunchKernelGGL((k_probe<0, 0>), dim3(wgs), dim3(256), 0, 0, d, 0); });
        float t1 = time_us([&] { hipLaunchKernelGGL((k_probe<36864, 0>), dim3(wgs), dim3(256), 0, 0, d, 0); });
        float t2 = time_us([&] { hipLaunchKernelGGL((k_probe<36864, 2500>), dim3(wgs), dim3(256), 0, 0, d, 0); });   // ~10k cycles = 4 us of work per wave
        float t3 = time_us([&] { hipLaunchKernelGGL((k_probe<0, 2500>), dim3(wgs), dim3(256), 0, 0, d, 0); });
        std::printf("%5d workgroups x 256 threads: empty %.1f us | 36 KB LDS %.1f us | 36 KB LDS + 4 us of work %.1f us | no LDS + 4 us of work %.1f us\n", wgs, t0, t1, t2, t3);
    }
    return 0;
}
