// mlp_stripe.h -- the PPO minibatch's forward and input-gradient chains as ONE launch each (both networks), row stripe by row stripe.
//
// What it replaces: per network 4 forward GEMM launches (k_gemm_nt, bias + ReLU epilogues) and 3 dX GEMM launches (ReLU-mask epilogue) of
// rlgpu_ppo_minibatch's bf16 path -- 14 dependent launches per minibatch whose every layer boundary went through HBM.  Reference semantics are
// unchanged: DiscretePolicy / ValueEstimator forward (PRIV/PPO/DiscretePolicy.h:27-31, ValueEstimator.cpp:10-23: Linear -> ReLU ... -> Linear)
// and autograd's backward through it (PPOLearner.cpp:205-215).
//
// A workgroup (4 wavefronts) owns a stripe of 32 * RT rows.  The stripe's activations stay in LDS from layer to layer (two ping-pong buffers,
// rows padded to 264 bf16 so that the 16 rows of a ds_read_b128 group land on distinct bank quads); every layer is
//     out[rows][N] = in[rows][K] . B            v_mfma_f32_32x32x16_bf16, fp32 accumulation,
// with the B operand -- the layer's weights, for dX its transpose -- streamed from L2 in FRAGMENT ORDER (k_weight_frags: the 64 lanes'
// 16-byte operands of one (32-column block, 16-deep K step) back to back, 1 KB per load, no LDS staging, no bank conflicts): all K steps of a
// wavefront's two column blocks are requested before the first MFMA.  What the dW GEMMs need afterwards -- the hidden activations, the
// activation gradients -- leaves for HBM with 16-byte stores while the next layer computes.  Operand values, accumulation order inside a
// 32x32 tile, bias / ReLU / bf16 rounding points are those of the per-layer kernels, so results are identical to them.
#pragma once

namespace stripe {

constexpr int MAXL = 4;          // layers per network the fused kernels take (the flagship has 4: 89 -> 256 -> 256 -> 256 -> 90 / 1)
constexpr int LD = 264;          // LDS activation row, elements (max width 256 + 8)
constexpr int MAXW = 256;        // widest layer

struct Net {
    int n_layers; int dims[MAXL + 1]; int kp[MAXL + 1];
    const short* wf[MAXL];       // forward B fragments: k_weight_frags of W   [N pad][kp[i]]
    const short* wtf[MAXL];      // dX B fragments:      k_weight_frags of W^T [K_in pad][kp[i+1]]
    const float* bias[MAXL];
    short* act[MAXL];            // act[i], i < L-1: hidden activation [rows][kp[i+1]] bf16 (written by fwd, read by bwd as ReLU mask and by the dW GEMMs)
    float* out32; int ld32;      // last layer: logits / values fp32 [rows][ld32]
    short* dy[MAXL];             // dy[i] = dL/d(layer i output) [rows][kp[i+1]] bf16; dy[L-1] comes from the loss kernel, the others are written by bwd
};
struct Args { const short* x16; int ldx; int rows; Net net[2]; };

using bf16x8_t = __attribute__((ext_vector_type(8))) short;
using f32x16_t = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ short f2bf_rn(float f) {   // the hardware conversion, exactly rlgpu_learn.hip's f2bf
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<short*>(&h);
}
__device__ __forceinline__ float bf2f_(short s) { return __uint_as_float(((unsigned int)(unsigned short)s) << 16); }

// acc[r][c] += in[32 r .. 32 r + 31][K] . B(column blocks cb0, cb0 + 1): every B fragment of the reduction in flight before the first MFMA
template <int NK, int RT>
__device__ __forceinline__ void block_mma(const short* in, const short* w0, bool two, int lane, f32x16_t (&acc)[RT][2]) {
    bf16x8_t b0[NK], b1[NK];
#pragma unroll
    for (int s = 0; s < NK; s++) b0[s] = *reinterpret_cast<const bf16x8_t*>(w0 + (size_t)s * 512);
    if (two) {
#pragma unroll
        for (int s = 0; s < NK; s++) b1[s] = *reinterpret_cast<const bf16x8_t*>(w0 + (size_t)(NK + s) * 512);
    }
    const short* arow = in + (lane & 31) * LD + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < NK; s++) {
        bf16x8_t a[RT];
#pragma unroll
        for (int r = 0; r < RT; r++) a[r] = *reinterpret_cast<const bf16x8_t*>(arow + r * 32 * LD + s * 16);
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b0[s], acc[r][0], 0, 0, 0);
        if (two) {
#pragma unroll
            for (int r = 0; r < RT; r++) acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[r], b1[s], acc[r][1], 0, 0, 0);
        }
    }
}
template <int RT>
__device__ __forceinline__ void block_mma_nk(int nk, const short* in, const short* w0, bool two, int lane, f32x16_t (&acc)[RT][2]) {
    switch (nk) {
        case 2: block_mma<2, RT>(in, w0, two, lane, acc); break;
        case 4: block_mma<4, RT>(in, w0, two, lane, acc); break;
        case 6: block_mma<6, RT>(in, w0, two, lane, acc); break;
        case 8: block_mma<8, RT>(in, w0, two, lane, acc); break;
        case 12: block_mma<12, RT>(in, w0, two, lane, acc); break;
        default: block_mma<16, RT>(in, w0, two, lane, acc); break;   // (host: nk is one of these)
    }
}
inline bool nk_supported(int nk) { return nk == 2 || nk == 4 || nk == 6 || nk == 8 || nk == 12 || nk == 16; }

// rows [m0, m0 + 32 RT) x cols [0, width) of a bf16 global matrix -> LDS (zeros beyond `rows`)
template <int RT>
__device__ __forceinline__ void load_stripe(short* dst, const short* src, int ld, int width, int m0, int rows, int tid) {
    const int cpr = width / 8;
    for (int idx = tid; idx < 32 * RT * cpr; idx += 256) {
        const int row = idx / cpr, ch = idx % cpr;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m0 + row < rows) v = *reinterpret_cast<const uint4*>(src + (size_t)(m0 + row) * ld + ch * 8);
        *reinterpret_cast<uint4*>(&dst[row * LD + ch * 8]) = v;
    }
}

// ---- forward: x16 -> act[0..L-2] (bf16, HBM + LDS) -> out32 ------------------------------------------------------------------------------
template <int RT>
__global__ void __launch_bounds__(256) k_fwd_stripe(Args g) {
    extern __shared__ __attribute__((aligned(16))) short st_smem[];
    const Net& n = g.net[blockIdx.y];
    if (n.n_layers == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 32 * RT;
    short* in = st_smem;
    short* out = st_smem + 32 * RT * LD;
    load_stripe<RT>(in, g.x16, g.ldx, n.kp[0], m0, g.rows, tid);
    __syncthreads();
    for (int i = 0; i < n.n_layers; i++) {
        const bool last = (i == n.n_layers - 1);
        const int nk = n.kp[i] / 16, N = n.dims[i + 1];
        const int nblk = last ? (N + 31) / 32 : n.kp[i + 1] / 32;
        for (int cb0 = wave * 2; cb0 < nblk; cb0 += 8) {
            const bool two = cb0 + 1 < nblk;
            f32x16_t acc[RT][2];
#pragma unroll
            for (int r = 0; r < RT; r++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[r][c][q] = 0.f;
            block_mma_nk<RT>(nk, in, n.wf[i] + ((size_t)cb0 * nk * 64 + lane) * 8, two, lane, acc);
            // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (c == 1 && !two) break;
                const int col = (cb0 + c) * 32 + (lane & 31);
                const float bias = (col < N) ? n.bias[i][col] : 0.f;
#pragma unroll
                for (int r = 0; r < RT; r++)
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int row = r * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                        const float v = acc[r][c][q] + bias;
                        if (last) { if (col < N && m0 + row < g.rows) n.out32[(size_t)(m0 + row) * n.ld32 + col] = v; }
                        else out[row * LD + col] = (col < N) ? f2bf_rn(fmaxf(v, 0.f)) : (short)0;
                    }
            }
        }
        if (last) break;
        __syncthreads();
        {   // the hidden activation leaves for HBM (the dW GEMM's operand, the backward chain's ReLU mask): 16 bytes per lane
            const int cpr = n.kp[i + 1] / 8;
            short* const dst = n.act[i];
            for (int idx = tid; idx < 32 * RT * cpr; idx += 256) {
                const int row = idx / cpr, ch = idx % cpr;
                if (m0 + row < g.rows) *reinterpret_cast<uint4*>(dst + (size_t)(m0 + row) * n.kp[i + 1] + ch * 8) = *reinterpret_cast<const uint4*>(&out[row * LD + ch * 8]);
            }
        }
        short* t = in; in = out; out = t;   // (the next layer writes the OTHER buffer: what was just read by the copy above stays intact)
    }
}

// ---- backward: dy[L-1] (from the loss kernel) -> dy[L-2] ... dy[0] -------------------------------------------------------------------------
// dX_i = (dY_i . W_i) masked by act[i-1] > 0, i = L-1 .. 1 (layer 0's input gradient is not needed)
template <int RT>
__global__ void __launch_bounds__(256) k_bwd_stripe(Args g) {
    extern __shared__ __attribute__((aligned(16))) short st_smem[];
    const Net& n = g.net[blockIdx.y];
    if (n.n_layers < 2) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * 32 * RT;
    short* in = st_smem;
    short* out = st_smem + 32 * RT * LD;
    const int L = n.n_layers;
    load_stripe<RT>(in, n.dy[L - 1], n.kp[L], n.kp[L], m0, g.rows, tid);
    __syncthreads();
    for (int i = L - 1; i >= 1; i--) {
        const int nk = n.kp[i + 1] / 16;           // reduction over the layer's outputs
        const int nblk = n.kp[i] / 32;             // columns = the layer's inputs
        for (int cb0 = wave * 2; cb0 < nblk; cb0 += 8) {
            const bool two = cb0 + 1 < nblk;
            f32x16_t acc[RT][2];
#pragma unroll
            for (int r = 0; r < RT; r++)
#pragma unroll
                for (int c = 0; c < 2; c++)
#pragma unroll
                    for (int q = 0; q < 16; q++) acc[r][c][q] = 0.f;
            block_mma_nk<RT>(nk, in, n.wtf[i] + ((size_t)cb0 * nk * 64 + lane) * 8, two, lane, acc);
#pragma unroll
            for (int c = 0; c < 2; c++) {
                if (c == 1 && !two) break;
                const int col = (cb0 + c) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < RT; r++)
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int row = r * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                        out[row * LD + col] = f2bf_rn(acc[r][c][q]);
                    }
            }
        }
        __syncthreads();
        {   // ReLU mask (the forward pass's activation, 16 bytes per lane from HBM), then to HBM for the dW GEMM and back into LDS for the next layer
            const int w = n.kp[i], cpr = w / 8, N = n.dims[i];
            const short* const mask = n.act[i - 1];
            short* const dst = n.dy[i - 1];
            for (int idx = tid; idx < 32 * RT * cpr; idx += 256) {
                const int row = idx / cpr, ch = idx % cpr;
                bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(&out[row * LD + ch * 8]);
                if (m0 + row < g.rows) {
                    const bf16x8_t mk = *reinterpret_cast<const bf16x8_t*>(mask + (size_t)(m0 + row) * w + ch * 8);
#pragma unroll
                    for (int q = 0; q < 8; q++) if (!(bf2f_(mk[q]) > 0.f) || ch * 8 + q >= N) v[q] = 0;
                    *reinterpret_cast<bf16x8_t*>(dst + (size_t)(m0 + row) * w + ch * 8) = v;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = 0;
                }
                if (i > 1) *reinterpret_cast<bf16x8_t*>(&out[row * LD + ch * 8]) = v;
            }
        }
        if (i > 1) __syncthreads();
        short* t = in; in = out; out = t;
    }
}

}  // namespace stripe
