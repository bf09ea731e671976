// rlgpu_learn.hip — the learner half of include/rlgpu.h on gfx950: DiscretePolicy / ValueEstimator MLPs on MFMA,
// fused softmax+sampler, GAE scan, fused PPO loss + d(logits), backward GEMMs, fused clip-by-norm + Adam.
//
// Reference semantics restated (file:line under RLGymPPO_CPP/src/private/RLGymPPO_CPP unless noted):
//   MLP           Linear -> ReLU (xk) -> Linear                       PPO/DiscretePolicy.cpp:11-24, ValueEstimator.cpp:10-23
//   probs         clamp(softmax(logits / T), 1e-11, 1), not renormalised  PPO/DiscretePolicy.h:27-32, .cpp:44-49
//   sampling      multinomial(probs, 1, true) == argmax(p / q), q~Exp(1)   PPO/DiscretePolicy.cpp:58-60
//   GAE           Util/TorchFuncs.cpp:5-52
//   PPO loss      PPO/PPOLearner.cpp:139-215
//   clip + Adam   PPO/PPOLearner.cpp:273-288, torch::optim::Adam defaults
//
// GEMMs use v_mfma_f32_32x32x2_f32 (exact fp32, the reference's precision) or, with use_bf16 (the reference's
// autocast dtype, FrameworkTorch.h:12-16), v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 master weights.
#include <hip/hip_runtime.h>
#include <sstream>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>

#include "../../include/rlgpu.h"
#include "rl_math.h"
#include "infer_device.h"
#include "rlgpu_internal.h"
#include "mlp_stripe.h"
#include "ppo_fused.h"

namespace {
using rlinfer::HeadArgs;
using rlinfer::policy_head_rows;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int GEMM_THREADS = 256;

struct GemmArgs {
    const float* A; int lda; int a_trans;   // a(m,k) = a_trans ? A[k*lda+m] : A[m*lda+k]
    const float* B; int ldb; int b_trans;   // b(k,n) = b_trans ? B[k*ldb+n] : B[n*ldb+k]
    float* C; int ldc;
    int M, N, K;
    const float* bias;                      // per column n, optional
    int relu;
    const float* mask; int ldmask;          // optional: C *= (mask(m,n) > 0)
    int atomic_accumulate;                  // C += result with atomics (split-K over blockIdx.z)
    int k_chunk;                            // K range per blockIdx.z
};

__device__ __forceinline__ short f2bf(float f) {
    // round-to-nearest-even; NaN stays NaN via the hardware conversion (MI355X_MICROARCH.md correctness table)
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<short*>(&h);
}

__device__ __forceinline__ float bf2f(short s) { return __uint_as_float(((uint32_t)(uint16_t)s) << 16); }

// C[M x N] = A[M x K] . B[K x N] on MFMA. 4 waves per block, each wave owns 32 rows x 64 cols of the 128 x 64 tile.
template <bool BF16>
__global__ void __launch_bounds__(GEMM_THREADS) k_gemm(GemmArgs g) {
    constexpr int LDA_S = BF16 ? (BK + 8) : (BK + 1);
    __shared__ __attribute__((aligned(16))) unsigned char smem[(BM + BN) * (BF16 ? (BK + 8) * 2 : (BK + 1) * 4)];
    float* As_f = reinterpret_cast<float*>(smem);
    float* Bs_f = As_f + BM * LDA_S;
    short* As_h = reinterpret_cast<short*>(smem);
    short* Bs_h = As_h + BM * LDA_S;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_chunk;
    const int kend = min(g.K, kbeg + g.k_chunk);

    f32x16 acc0, acc1;
    for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }

    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        // ---- stage A tile (BM x BK) and B tile (BN x BK) as [row][k]
        if (!g.a_trans) {
            for (int i = tid; i < BM * BK; i += GEMM_THREADS) {
                int r = i / BK, k = i % BK;
                int gm = m0 + r, gk = k0 + k;
                float v = (gm < g.M && gk < kend) ? g.A[(size_t)gm * g.lda + gk] : 0.f;
                if (BF16) As_h[r * LDA_S + k] = f2bf(v); else As_f[r * LDA_S + k] = v;
            }
        } else {
            for (int i = tid; i < BM * BK; i += GEMM_THREADS) {
                int k = i / BM, r = i % BM;
                int gm = m0 + r, gk = k0 + k;
                float v = (gm < g.M && gk < kend) ? g.A[(size_t)gk * g.lda + gm] : 0.f;
                if (BF16) As_h[r * LDA_S + k] = f2bf(v); else As_f[r * LDA_S + k] = v;
            }
        }
        if (!g.b_trans) {
            for (int i = tid; i < BN * BK; i += GEMM_THREADS) {
                int r = i / BK, k = i % BK;
                int gn = n0 + r, gk = k0 + k;
                float v = (gn < g.N && gk < kend) ? g.B[(size_t)gn * g.ldb + gk] : 0.f;
                if (BF16) Bs_h[r * LDA_S + k] = f2bf(v); else Bs_f[r * LDA_S + k] = v;
            }
        } else {
            for (int i = tid; i < BN * BK; i += GEMM_THREADS) {
                int k = i / BN, r = i % BN;
                int gn = n0 + r, gk = k0 + k;
                float v = (gn < g.N && gk < kend) ? g.B[(size_t)gk * g.ldb + gn] : 0.f;
                if (BF16) Bs_h[r * LDA_S + k] = f2bf(v); else Bs_f[r * LDA_S + k] = v;
            }
        }
        __syncthreads();
        const int arow = wave * 32 + (lane & 31);
        if (BF16) {
            // v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31], j = 0..7
#pragma unroll
            for (int ks = 0; ks < BK; ks += 16) {
                int kk = ks + 8 * (lane >> 5);
                bf16x8 a = *reinterpret_cast<const bf16x8*>(&As_h[arow * LDA_S + kk]);
                bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs_h[(lane & 31) * LDA_S + kk]);
                bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs_h[(32 + (lane & 31)) * LDA_S + kk]);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
            }
        } else {
            // v_mfma_f32_32x32x2_f32: lane l holds A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]
#pragma unroll
            for (int ks = 0; ks < BK; ks += 2) {
                int kk = ks + (lane >> 5);
                float a = As_f[arow * LDA_S + kk];
                float b0 = Bs_f[(lane & 31) * LDA_S + kk];
                float b1 = Bs_f[(32 + (lane & 31)) * LDA_S + kk];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // ---- epilogue. C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const f32x16& acc = t == 0 ? acc0 : acc1;
        int gn = n0 + t * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; r++) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            int gm = m0 + wave * 32 + row;
            if (gm < g.M && gn < g.N) {
                float v = acc[r];
                if (g.bias && blockIdx.z == 0) v += g.bias[gn];
                if (g.relu) v = fmaxf(v, 0.f);
                if (g.mask) v = (g.mask[(size_t)gm * g.ldmask + gn] > 0.f) ? v : 0.f;
                float* dst = &g.C[(size_t)gm * g.ldc + gn];
                if (g.atomic_accumulate) atomicAdd(dst, v); else *dst = v;
            }
        }
    }
}

// =====================================================================================================================
// bf16 fast path (use_bf16): activations, their gradients and shadow copies of the weights live in HBM as bf16, padded so
// that every reduction dimension is a multiple of 32 and every operand row is 16-byte aligned; fp32 master weights, fp32
// accumulation, fp32 logits / values / dW.  Two GEMM kernels cover the MLP:
//   k_gemm_nt  C[M x N]  = A[M x K] . B[N x K]^T   both operands K-contiguous (forward with W, dX with the W^T shadow)
//   k_gemm_tn  dW[Mo x No] += Y[R x Mo]^T . X[R x No]  both operands row(R)-major: the reduction runs DOWN the rows, so the
//              MFMA operands are gathered with ds_read_b64_tr_b16 (gfx950 transposing LDS read, cdna_hip_programming.md T10)
// =====================================================================================================================
using bf16x4 = __attribute__((ext_vector_type(4))) short;
constexpr int FBK = 32;          // K step of both kernels
constexpr int NT_LD = FBK + 8;   // LDS row of the NT tiles: 80 B -> the 16 rows of a ds_read_b128 group land on distinct bank quads
#ifndef RLG_NT_BK
#define RLG_NT_BK 64
#endif
constexpr int NBK = RLG_NT_BK;   // K step of k_gemm_nt: 64 -> every row piece a wavefront loads is a full 128-byte line (with 32 it was half of one): 0.55 -> 0.517 ms per minibatch; 128 (69 KB of LDS, 2 workgroups per CU) 0.559
constexpr int NTK_LD = NBK + 8;  // 144 B rows: the 16 rows of a ds_read_b128 group start 36 banks apart -> distinct bank quads

struct NtArgs {
    const short* A; int lda;     // [M][lda] bf16, lda = K (multiple of 32), zero padded
    const short* B; int ldb;     // [>= gridDim.x * BN][ldb] bf16, zero padded rows and columns
    int M, N, K;
    const float* bias;           // EPI 0/1
    short* C16; int ldc16;       // EPI 0/2: bf16 out, columns N..ldc16 written as zeros
    float* C32; int ldc32;       // EPI 1
    const short* mask16; int ldm;  // EPI 2: C = mask > 0 ? acc : 0
};

// EPI 0: C16 = relu(acc + bias)   EPI 1: C32 = acc + bias   EPI 2: C16 = mask16 > 0 ? acc : 0
template <int WM, int WN, int TM, int TN, int EPI>
__global__ void __launch_bounds__(256) k_gemm_nt(NtArgs g) {
    static_assert(WM * WN == 4, "4 wavefronts per workgroup");
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int CPR = NBK / 8;                                                    // 16-byte chunks per tile row
    constexpr int A_CH = (BM * CPR + 255) / 256, B_CH = (BN * CPR + 255) / 256;   // 16-byte chunks per thread and tile
    // bf16 outputs of the 128-wide configuration leave through LDS: the MFMA result layout has one column per lane (2-byte stores,
    // 2-byte mask loads); staged, every lane moves 16 contiguous bytes
    constexpr bool STAGED = (EPI != 1) && (BN == 128);
    constexpr int C_LD = BN + 8;
    constexpr int SMEM = STAGED ? ((BM + BN) * NTK_LD > BM * C_LD ? (BM + BN) * NTK_LD : BM * C_LD) : (BM + BN) * NTK_LD;
    __shared__ __attribute__((aligned(16))) short smem[SMEM];
    short* const As = smem;
    short* const Bs = smem + BM * NTK_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    uint4 ra[A_CH], rb[B_CH];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int j = 0; j < A_CH; j++) {
            int idx = tid + j * 256, row = idx / CPR, ch = idx % CPR;
            ra[j] = make_uint4(0, 0, 0, 0);
            if (idx < BM * CPR && m0 + row < g.M && k0 + ch * 8 < g.K) ra[j] = *reinterpret_cast<const uint4*>(g.A + (size_t)(m0 + row) * g.lda + k0 + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < B_CH; j++) {
            int idx = tid + j * 256, row = idx / CPR, ch = idx % CPR;
            rb[j] = make_uint4(0, 0, 0, 0);
            if (idx < BN * CPR && k0 + ch * 8 < g.K) rb[j] = *reinterpret_cast<const uint4*>(g.B + (size_t)(n0 + row) * g.ldb + k0 + ch * 8);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < A_CH; j++) {
            int idx = tid + j * 256, row = idx / CPR, ch = idx % CPR;
            if (idx < BM * CPR) *reinterpret_cast<uint4*>(&As[row * NTK_LD + ch * 8]) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < B_CH; j++) {
            int idx = tid + j * 256, row = idx / CPR, ch = idx % CPR;
            if (idx < BN * CPR) *reinterpret_cast<uint4*>(&Bs[row * NTK_LD + ch * 8]) = rb[j];
        }
    };

    load_tiles(0);
    for (int k0 = 0; k0 < g.K; k0 += NBK) {
        store_tiles();
        __syncthreads();
        if (k0 + NBK < g.K) load_tiles(k0 + NBK);   // next tile's global loads fly under this tile's MFMAs
#pragma unroll
        for (int ks = 0; ks < NBK; ks += 16) {
            // v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31], j = 0..7
            const int kk = ks + 8 * (lane >> 5);
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const bf16x8*>(&As[((wm * TM + i) * 32 + (lane & 31)) * NTK_LD + kk]);
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = *reinterpret_cast<const bf16x8*>(&Bs[((wn * TN + j) * 32 + (lane & 31)) * NTK_LD + kk]);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (STAGED) {
        short* const Cs = smem;   // the k-loop ended with a barrier: As / Bs are dead
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++) {
                const int col = (wn * TN + j) * 32 + (lane & 31);
                float bias = 0.f;
                if (EPI == 0 && g.bias && n0 + col < g.N) bias = g.bias[n0 + col];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    float v = acc[i][j][r] + bias;
                    if (EPI == 0) v = fmaxf(v, 0.f);
                    Cs[row * C_LD + col] = f2bf(v);
                }
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (BM * (BN / 8)) / 256; it++) {
            const int idx = tid + it * 256, row = idx / (BN / 8), ch = idx % (BN / 8);
            const int gm = m0 + row, gn = n0 + ch * 8;
            if (gm >= g.M || gn >= g.ldc16) continue;
            bf16x8 v = *reinterpret_cast<const bf16x8*>(&Cs[row * C_LD + ch * 8]);
            if (EPI == 2) {
                const bf16x8 mk = *reinterpret_cast<const bf16x8*>(&g.mask16[(size_t)gm * g.ldm + gn]);
#pragma unroll
                for (int q = 0; q < 8; q++) if (!(bf2f(mk[q]) > 0.f)) v[q] = 0;
            }
            if (gn + 8 > g.N) {
#pragma unroll
                for (int q = 0; q < 8; q++) if (gn + q >= g.N) v[q] = 0;
            }
            *reinterpret_cast<bf16x8*>(&g.C16[(size_t)gm * g.ldc16 + gn]) = v;
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int gn = n0 + (wn * TN + j) * 32 + (lane & 31);
            float bias = 0.f;
            if (EPI != 2 && g.bias && gn < g.N) bias = g.bias[gn];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int gm = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm >= g.M) continue;
                float v = acc[i][j][r] + bias;
                if (EPI == 0) {
                    if (gn < g.ldc16) g.C16[(size_t)gm * g.ldc16 + gn] = (gn < g.N) ? f2bf(fmaxf(v, 0.f)) : (short)0;
                } else if (EPI == 1) {
                    if (gn < g.N) g.C32[(size_t)gm * g.ldc32 + gn] = v;
                } else {
                    if (gn < g.ldc16) {
                        short o = 0;
                        if (gn < g.N && bf2f(g.mask16[(size_t)gm * g.ldm + gn]) > 0.f) o = f2bf(v);
                        g.C16[(size_t)gm * g.ldc16 + gn] = o;
                    }
                }
            }
        }
}

struct TnArgs {
    const short* Y; int ldy;     // [R][ldy] bf16: dL/d(layer output), columns Mo..ldy zero
    const short* X; int ldx;     // [R][ldx] bf16: layer input, columns No..ldx zero
    int R, Mo, No;
    float* dW; int ldw;          // [Mo][ldw] fp32, += with atomics
    float* db;                   // [Mo] fp32, += column sums of Y
    int slab;                    // rows per blockIdx.z
};
constexpr int TN_LD = 160;       // LDS row (elements) of the 128-wide k-major tiles: 320 B = 16 banks (mod 64) per k row, so
                                 // the 4 rows x 4 column-quads x 2 groups of a transposing read hit 64 distinct banks

__device__ __forceinline__ bf16x8 tr_operand(const short* S, int k16, int col0, int lane) {
    // 32 (cols) x 16 (k) MFMA operand out of a k-major LDS tile: per 16-lane group a 4(k) x 16(col) block, delivered
    // column-major; lane 4q+p of the group supplies the address of block row q, columns 4p..4p+3
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int kb = k16 + 8 * (g >> 1), cb = col0 + 16 * (g & 1) + 4 * p;
    using lds_v4 = __attribute__((address_space(3))) bf16x4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(S + (kb + q) * TN_LD + cb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(S + (kb + 4 + q) * TN_LD + cb));
    bf16x8 r; r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

__global__ void __launch_bounds__(256) k_gemm_tn(TnArgs g) {
    __shared__ __attribute__((aligned(16))) short Ys[FBK * TN_LD];
    __shared__ __attribute__((aligned(16))) short Xs[FBK * TN_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int r_begin = blockIdx.z * g.slab, r_end = min(g.R, r_begin + g.slab);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
    float bsum = 0.f;
    const bool do_bias = (blockIdx.x == 0) && g.db;

    uint4 ry[2], rx[2];
    auto load_tiles = [&](int r0) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int idx = tid + j * 256, row = idx >> 4, ch = idx & 15;
            int gr = r0 + row, cy = m0 + ch * 8, cx = n0 + ch * 8;
            ry[j] = make_uint4(0, 0, 0, 0); rx[j] = make_uint4(0, 0, 0, 0);
            if (gr < r_end) {
                if (cy < g.ldy) ry[j] = *reinterpret_cast<const uint4*>(g.Y + (size_t)gr * g.ldy + cy);
                if (cx < g.ldx) rx[j] = *reinterpret_cast<const uint4*>(g.X + (size_t)gr * g.ldx + cx);
            }
        }
    };
    load_tiles(r_begin);
    for (int r0 = r_begin; r0 < r_end; r0 += FBK) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
            int idx = tid + j * 256, row = idx >> 4, ch = idx & 15;
            *reinterpret_cast<uint4*>(&Ys[row * TN_LD + ch * 8]) = ry[j];
            *reinterpret_cast<uint4*>(&Xs[row * TN_LD + ch * 8]) = rx[j];
        }
        __syncthreads();
        if (r0 + FBK < r_end) load_tiles(r0 + FBK);
        if (do_bias && tid < 128) {
#pragma unroll 8
            for (int k = 0; k < FBK; k++) bsum += bf2f(Ys[k * TN_LD + tid]);
        }
#pragma unroll
        for (int ks = 0; ks < FBK; ks += 16) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; i++) a[i] = tr_operand(Ys, ks, (wm * 2 + i) * 32, lane);
#pragma unroll
            for (int j = 0; j < 2; j++) b[j] = tr_operand(Xs, ks, (wn * 2 + j) * 32, lane);
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int gn = n0 + (wn * 2 + j) * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int gm = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm < g.Mo && gn < g.No) atomicAdd(&g.dW[(size_t)gm * g.ldw + gn], acc[i][j][r]);
            }
        }
    if (do_bias && tid < 128 && m0 + tid < g.Mo) atomicAdd(&g.db[m0 + tid], bsum);
}

// fp32 rows (optionally gathered through idx) -> bf16 rows padded with zeros to ld16 columns
__global__ void k_rows_to_bf16(const float* src, const int32_t* idx, int rows, int D, short* dst, int ld16) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * ld16) return;
    int r = (int)(i / ld16), c = (int)(i % ld16);
    float v = 0.f;
    if (c < D) v = src[(size_t)(idx ? idx[r] : r) * D + c];
    dst[i] = f2bf(v);
}

// bf16 shadows of one Linear layer's fp32 master weight W[N][K]:  w16[rows16][ld16] = W (zero padded),  wt16[rowst][ldt] = W^T
template <bool HALF>
__global__ void k_weight_shadows(const float* W, int N, int K, short* w16, int rows16, int ld16, short* wt16, int rowst, int ldt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n1 = (size_t)rows16 * ld16, n2 = (size_t)rowst * ldt;
    if (i < n1) {
        int r = (int)(i / ld16), c = (int)(i % ld16);
        w16[i] = (r < N && c < K) ? fused::f2s<HALF>(W[(size_t)r * K + c]) : (short)0;
    } else if (i < n1 + n2) {
        size_t j = i - n1;
        int r = (int)(j / ldt), c = (int)(j % ldt);
        wt16[j] = (r < K && c < N) ? fused::f2s<HALF>(W[(size_t)c * K + r]) : (short)0;
    }
}

// w16 [rows16][K] -> the order k_mlp_infer consumes it in: for every 32-row block cb and K step s (16 deep) the 64 lanes' 16-byte
// MFMA B operands back to back (1 KB, one fully coalesced load):  wf[((cb * K/16 + s) * 64 + lane) * 8 + j] = W[cb*32 + (lane&31)][s*16 + 8*(lane>>5) + j]
__global__ void k_weight_frags(const short* w16, int rows16, int K, short* wf) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows16 * K) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const size_t blk = i >> 9;
    const int nk = K / 16, s = (int)(blk % nk), cb = (int)(blk / nk);
    wf[i] = w16[(size_t)(cb * 32 + (lane & 31)) * K + s * 16 + 8 * (lane >> 5) + j];
}

// All four 16-bit copies of every Linear layer of both networks in ONE launch (was k_weight_shadows + 2 x k_weight_frags per layer: 24 launches per
// optimizer step): blockIdx.y = layer; thread i writes element i of the row-major copies and element i of the fragment-ordered ones.
struct RefreshLayer { const float* W; int N, K; short *w16, *wt16, *wf, *wtf; int rows16, ld16, rowst, ldt; };
struct RefreshArgs { RefreshLayer L[18]; };
template <bool HALF>
__global__ void k_refresh_all(RefreshArgs g) {
    const RefreshLayer& L = g.L[blockIdx.y];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n1 = (size_t)L.rows16 * L.ld16, n2 = (size_t)L.rowst * L.ldt;
    auto w_at = [&](int r, int c) { return (r < L.N && c < L.K) ? fused::f2s<HALF>(L.W[(size_t)r * L.K + c]) : (short)0; };
    auto frag_rc = [](size_t idx, int K, int& r, int& c) {    // k_weight_frags: fragment element idx of a [rows][K] matrix sits at (r, c)
        const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63); const size_t blk = idx >> 9; const int nk = K / 16, s = (int)(blk % nk), cb = (int)(blk / nk);
        r = cb * 32 + (lane & 31); c = s * 16 + 8 * (lane >> 5) + j;
    };
    if (i < n1) {
        L.w16[i] = w_at((int)(i / L.ld16), (int)(i % L.ld16));
        int r, c; frag_rc(i, L.ld16, r, c); L.wf[i] = w_at(r, c);
    }
    if (i < n2) {
        L.wt16[i] = w_at((int)(i % L.ldt), (int)(i / L.ldt));                 // W^T[r][c] = W[c][r]
        int r, c; frag_rc(i, L.ldt, r, c); L.wtf[i] = w_at(c, r);
    }
}

// column sums: out[n] += sum_m X[m][n]  (fp32 path; the bf16 path folds them into k_gemm_tn)
__global__ void __launch_bounds__(256) k_col_sum(const float* X, int ld, int M, int N, float* out) {
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float s = 0.f;
    if (n < N) for (int m = m0 + rg; m < m1; m += 4) s += X[(size_t)m * ld + n];
    part[rg][c] = s;
    __syncthreads();
    if (rg == 0 && n < N) atomicAdd(&out[n], part[0][c] + part[1][c] + part[2][c] + part[3][c]);
}

__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

// two rows per wavefront (32 lanes each): probs = clamp(softmax(logits / T), 1e-11, 1); action = argmax(p / q) or argmax(p); logp = log p[a]  (HeadArgs,
// policy_head_rows: infer_device.h); an odd last row is done by both halves
__global__ void k_policy_head(const float* logits, int ld, int rows, HeadArgs h) {
    const int r0 = 2 * (blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    if (r0 >= rows) return;
    const int r1 = r0 + 1 < rows ? r0 + 1 : r0;
    const float* const zs[2] = {logits + (size_t)r0 * ld, logits + (size_t)r1 * ld}; const int rw[2] = {r0, r1};
    policy_head_rows<2>(zs, rw, lane, h);
}

// ---- fused MLP inference (bf16 path): obs -> [Linear+ReLU]* -> Linear -> head, ONE launch --------------------------------
// Collection calls the policy once per gym step on n_agents rows (8192): as separate kernels that is input staging + one GEMM per
// layer + the head, ~45 us of kernels and as much again in launch gaps next to a ~670 us env step.  Here a workgroup owns 32 rows:
// their activations never leave LDS (two ping-pong buffers), each of the 4 wavefronts computes 32-column blocks of the layer
// output, taking the A operand from LDS and the B operand (the bf16 weight shadow, K-contiguous rows) straight from L2 with one
// 16-byte load per lane and K step.  Same operand values, accumulation order, bias / ReLU / rounding as k_gemm_nt, so the logits
// are those of the unfused path; the head is the same code (policy_head_row) reading the logits from LDS.
constexpr int FI_ROWS = 32;          // rows per workgroup
constexpr int FI_WAVES = 8;          // wavefronts per workgroup: one 32-column block of a 256-wide layer each
constexpr int FI_LOGIT_LD = 132;     // fp32 logits row in LDS (n_actions <= 128)
constexpr int FI_CHUNK = 16;         // K steps (of 16) whose B operands are in flight together: a whole 256-deep reduction
struct FusedInferArgs {
    const float* obs; int D, rows, n_layers;
    const short* W[9]; const float* bias[9]; int K[9], N[9], Npad[9];   // layer i: W = the fragment-ordered weight shadow (k_weight_frags), K = kp[i], N = dims[i+1], Npad = kp[i+1]
    int ld;                          // LDS activation row (elements): max kp + 8
    int buf_elems;                   // elements (shorts) per ping-pong buffer
    int mode;                        // 0: policy head, 1: values[row] = output column 0
    float* values;
    HeadArgs head;
    unsigned long long* stamps;      // debug: phase clock stamps of workgroup 0 / wavefront 0 (NULL in production)
};

// The kernel is a chain of short dependent phases (a layer is 16 MFMAs per wavefront), so what matters is latency: every
// wavefront asks for the weights of its NEXT column block before it starts the MFMAs of the current one -- they do not depend on
// the activations -- and the loads land under the MFMAs, the epilogue and the barrier.
__global__ void __launch_bounds__(64 * FI_WAVES) k_mlp_infer(FusedInferArgs g) {
    extern __shared__ __attribute__((aligned(16))) short fi_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * FI_ROWS;
    short* in = fi_smem;
    short* out = fi_smem + g.buf_elems;
    auto n_blocks = [&](int i) { return (i == g.n_layers - 1) ? (g.N[i] + 31) / 32 : g.Npad[i] / 32; };
    // v_mfma_f32_32x32x16_bf16: lane l holds A[row l&31][k = 8*(l>>5) + j], B[k = 8*(l>>5) + j][col l&31], j = 0..7
    auto fetch = [&](bf16x8 (&b)[FI_CHUNK], int i, int cb, int s0) {
        const int nk = g.K[i] / 16;
        const short* w = g.W[i] + ((size_t)cb * nk * 64 + lane) * 8;   // fragment order (k_weight_frags): 1 KB per (block, K step)
        if (s0 == 0 && nk <= FI_CHUNK) {
            // guard-free code per depth (infer_device.h): behind per-step guards the compiler waits for every outstanding load in
            // front of every MFMA, and the prefetch of the next layer's weights stalls the current layer
#define RLINFER_FETCH(NK) rlinfer::fetch_block<NK>(b, w)
            RLINFER_DISPATCH_NK(nk, RLINFER_FETCH, rlinfer::fetch_block_any(b, w, nk))
#undef RLINFER_FETCH
            return;
        }
#pragma unroll
        for (int j = 0; j < FI_CHUNK; j++)
            if (s0 + j < nk) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)(s0 + j) * 512);
    };
    int n_stamp = 0;
    auto stamp = [&]() { if (g.stamps && blockIdx.x == 0 && tid == 0) g.stamps[n_stamp++] = __builtin_amdgcn_s_memtime(); };
    stamp();
    bf16x8 bnext[FI_CHUNK];
    if (wave < n_blocks(0)) fetch(bnext, 0, wave, 0);
    // fp32 observations -> bf16, zero padded to K[0] columns (k_rows_to_bf16)
    for (int r = wave; r < FI_ROWS; r += FI_WAVES)
        for (int c = lane; c < g.K[0]; c += 64) {
            float v = 0.f;
            if (m0 + r < g.rows && c < g.D) v = g.obs[(size_t)(m0 + r) * g.D + c];
            in[r * g.ld + c] = f2bf(v);
        }
    __syncthreads();
    stamp();
    for (int i = 0; i < g.n_layers; i++) {
        const bool last = (i == g.n_layers - 1);
        const int K = g.K[i], N = g.N[i], nk = K / 16, nblk = n_blocks(i);
        float* const logits = reinterpret_cast<float*>(out);
        const short* arow = in + (lane & 31) * g.ld + 8 * (lane >> 5);
        bool prefetched_next = false;
        for (int cb = wave; cb < nblk; cb += FI_WAVES) {
            bf16x8 b[FI_CHUNK];
            if (cb == wave) {
#pragma unroll
                for (int j = 0; j < FI_CHUNK; j++) b[j] = bnext[j];
            } else fetch(b, i, cb, 0);
            if (cb + FI_WAVES >= nblk && !last && wave < n_blocks(i + 1)) { fetch(bnext, i + 1, wave, 0); prefetched_next = true; }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
            if (nk <= FI_CHUNK) {
#define RLINFER_MMA(NK) rlinfer::mma_block<NK>(arow, b, acc)
                RLINFER_DISPATCH_NK(nk, RLINFER_MMA, rlinfer::mma_block_any(arow, b, acc, nk))
#undef RLINFER_MMA
            } else
            for (int s0 = 0; s0 < nk; s0 += FI_CHUNK) {
                if (s0 > 0) fetch(b, i, cb, s0);   // reductions deeper than 256: the rest arrives chunk by chunk
#pragma unroll
                for (int j = 0; j < FI_CHUNK; j++) {
                    if (s0 + j >= nk) break;
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + (s0 + j) * 16);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc, 0, 0, 0);
                }
            }
            // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
            const int col = cb * 32 + (lane & 31);
            const float bias = (col < N) ? g.bias[i][col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float v = acc[r] + bias;
                if (last) logits[row * FI_LOGIT_LD + col] = v;
                else out[row * g.ld + col] = (col < N) ? f2bf(fmaxf(v, 0.f)) : (short)0;
            }
        }
        if (!prefetched_next && !last && wave < n_blocks(i + 1)) fetch(bnext, i + 1, wave, 0);   // a wavefront without a block in this layer
        __syncthreads();
        stamp();
        short* t = in; in = out; out = t;
    }
    const float* logits = reinterpret_cast<const float*>(in);   // the last layer wrote into what is now `in`
    if (g.mode == 1) {
        if (tid < FI_ROWS && m0 + tid < g.rows) g.values[m0 + tid] = logits[tid * FI_LOGIT_LD];
        return;
    }
    {   // the wavefront's FI_ROWS / FI_WAVES rows together (rows past the end redo the last real one: same values, same stores)
        constexpr int NR = FI_ROWS / FI_WAVES;
        const float* zs[NR]; int rows[NR];
#pragma unroll
        for (int n = 0; n < NR; n++) {
            int r = wave + n * FI_WAVES;
            if (m0 + r >= g.rows) r = g.rows - 1 - m0;
            zs[n] = logits + r * FI_LOGIT_LD; rows[n] = m0 + r;
        }
        const float* const (&zc)[NR] = reinterpret_cast<const float* const (&)[NR]>(zs);
        policy_head_rows<NR>(zc, rows, lane, g.head);
    }
    stamp();
}

// GAE: one lane per agent trajectory (column j of the time-major arrays), reverse scan over T (TorchFuncs.cpp:23-43)
__global__ void k_gae(const float* rews, const float* dones, const float* truncs, const float* values, int T, int n,
                      float gamma, float lambda, float ret_std, float clip_range, int mode, float* adv, float* targets, float* returns) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    float last_gae = 0.f, last_ret = 0.f;
    for (int t = T - 1; t >= 0; t--) {
        size_t i = (size_t)t * n + j;
        float done = 1.f - dones[i];
        float trunc = 1.f - truncs[i];
        float nv;
        if (t == T - 1) {
            if (mode == 0) nv = (j == n - 1) ? values[(size_t)T * n + j] : values[j + 1];  // first state of the NEXT trajectory (Q1)
            else nv = values[(size_t)T * n + j];
        } else nv = values[(size_t)(t + 1) * n + j];
        float norm_rew;
        if (ret_std != 0.f) {
            norm_rew = rews[i] / ret_std;
            if (clip_range > 0.f) norm_rew = fminf(fmaxf(norm_rew, -clip_range), clip_range);
        } else norm_rew = rews[i];
        float pred_ret = norm_rew + gamma * nv * done;
        float delta = pred_ret - values[i];
        float ret = rews[i] + last_ret * gamma * done * trunc;
        returns[i] = ret;
        last_ret = ret;
        last_gae = delta + gamma * lambda * done * trunc * last_gae;
        adv[i] = last_gae;
        targets[i] = values[i] + last_gae;
    }
}

// The same scan over trajectories of DIFFERENT lengths (free-running collection, rlgpu_collect_free): agent j's trajectory has
// steps[j / players] rows, rows beyond it do not exist.  The batch the reference hands to ComputeGAE is the concatenation of the non-empty
// trajectories (ThreadAgentManager.cpp:47-60), so with mode 0 "the next row" after a trajectory's last step is the first state of the
// next NON-EMPTY trajectory, and the batch's very last row is followed by the value of its own next state (Learner.cpp:619-640).
// With `trunc_marks` null the collector's truncation mark is applied here: 1 - done on a trajectory's last row, 0 elsewhere (ThreadAgentManager.cpp:55).
__global__ void k_gae_ragged(const float* rews, const float* dones, const float* truncs, const float* values, int n, const int32_t* steps, int players,
                             float gamma, float lambda, float ret_std, float clip_range, int mode, float* adv, float* targets, float* returns) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int len = steps[j / players];
    if (len <= 0) return;
    float nv_last;
    if (mode == 0) {
        int j2 = j + 1;
        while (j2 < n && steps[j2 / players] <= 0) j2++;
        nv_last = j2 < n ? values[j2] : values[(size_t)len * n + j];
    } else nv_last = values[(size_t)len * n + j];
    float last_gae = 0.f, last_ret = 0.f;
    for (int t = len - 1; t >= 0; t--) {
        size_t i = (size_t)t * n + j;
        float done = 1.f - dones[i];
        float trunc = truncs ? 1.f - truncs[i] : (t == len - 1 ? dones[i] : 1.f);   // 1 - truncated; truncated = 1 - done on the last row
        float nv = (t == len - 1) ? nv_last : values[(size_t)(t + 1) * n + j];
        float norm_rew;
        if (ret_std != 0.f) {
            norm_rew = rews[i] / ret_std;
            if (clip_range > 0.f) norm_rew = fminf(fmaxf(norm_rew, -clip_range), clip_range);
        } else norm_rew = rews[i];
        float pred_ret = norm_rew + gamma * nv * done;
        float delta = pred_ret - values[i];
        float ret = rews[i] + last_ret * gamma * done * trunc;
        returns[i] = ret;
        last_ret = ret;
        last_gae = delta + gamma * lambda * done * trunc * last_gae;
        adv[i] = last_gae;
        targets[i] = values[i] + last_gae;
    }
}

// off[a] = rows of the trajectories before agent a's in the concatenated batch (exclusive prefix sum of steps[a / players]), off[n] = all rows.
// One workgroup of 1024 threads: contiguous ranges per thread, the 1024 range sums scanned in LDS.
__global__ void __launch_bounds__(1024) k_traj_offsets(const int32_t* steps, int n, int players, int32_t* off) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (n + 1023) / 1024, a0 = tid * per, a1 = min(n, a0 + per);
    int sum = 0;
    for (int a = a0; a < a1; a++) sum += max(steps[a / players], 0);
    part[tid] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) { int v = tid >= o ? part[tid - o] : 0; __syncthreads(); part[tid] += v; __syncthreads(); }
    int run = part[tid] - sum;
    for (int a = a0; a < a1; a++) { off[a] = run; run += max(steps[a / players], 0); }
    if (tid == 1023) off[n] = part[1023];
}

// GetAllBatchesShuffled's permutation (logical FIFO rows, oldest iteration first, trajectory after trajectory inside one) -> device rows
// slot * slot_rows + t * n + agent, for iterations whose trajectories have their own lengths: off_base + slot * (n + 1) = that slot's k_traj_offsets
struct ExpChunks { int count; int slot[16]; long long start[17]; long long skip[16]; };
__global__ void k_map_rows(const int32_t* perm, long long cur, ExpChunks ch, const int32_t* off_base, int n, long long slot_rows, int32_t* rows) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cur) return;
    const long long p = perm[i];
    int c = 0;
    while (c + 1 < ch.count && p >= ch.start[c + 1]) c++;
    const long long a = p - ch.start[c] + ch.skip[c];
    const int32_t* off = off_base + (size_t)ch.slot[c] * (n + 1);
    int lo = 0, hi = n;   // the last agent whose offset is <= a (empty trajectories share their successor's offset: take the one that owns row a)
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if ((long long)off[mid] <= a) lo = mid; else hi = mid; }
    rows[i] = (int32_t)((long long)ch.slot[c] * slot_rows + (a - off[lo]) * n + lo);
}

// fused PPO policy loss + gradient wrt logits; one wave per row, grid-stride over rows. (PPOLearner.cpp:148-198, DiscretePolicy.cpp:64-75)
// metrics: [0] entropy sum, [1] KL sum, [2] clip count, [3] ratio sum (per-row sums; host divides).  Each wave keeps its
// sums in registers, the workgroup folds them in LDS and issues ONE atomic per metric (65536 rows x 4 same-address atomics
// cost 3.3 ms before).  dlogits goes out as fp32 [rows][ld] or, when d16 is set, as bf16 [rows][ld16] (zero padded).
__global__ void __launch_bounds__(256) k_ppo_policy_loss(const float* logits, int ld, int rows, int A, float inv_temp, const int32_t* actions, const float* old_logp,
                                                         const float* adv, const int32_t* idx, float clip, float ent_coef, float scale /* ratio / rows */,
                                                         float* dlogits, short* d16, int ld16, float* metrics) {
    __shared__ float red[4][4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m_ent = 0.f, m_kl = 0.f, m_clip = 0.f, m_ratio = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        int src = idx ? idx[row] : row;
        const float* z = logits + (size_t)row * ld;
        float v0 = lane < A ? z[lane] * inv_temp : -INFINITY, v1 = (lane + 64) < A ? z[lane + 64] * inv_temp : -INFINITY;
        float mx = wave_max(fmaxf(v0, v1));
        float e0 = lane < A ? expf(v0 - mx) : 0.f, e1 = (lane + 64) < A ? expf(v1 - mx) : 0.f;
        float sum = wave_sum(e0 + e1);
        float s0 = e0 / sum, s1 = e1 / sum;
        float p0 = fminf(fmaxf(s0, 1e-11f), 1.f), p1 = fminf(fmaxf(s1, 1e-11f), 1.f);
        float lp0 = logf(p0), lp1 = logf(p1);
        float ent = wave_sum((lane < A ? -lp0 * p0 : 0.f) + ((lane + 64) < A ? -lp1 * p1 : 0.f));
        int a = actions[src];
        float logp_a = (a < 64) ? __shfl(lp0, a, 64) : __shfl(lp1, a - 64, 64);
        float p_a = (a < 64) ? __shfl(p0, a, 64) : __shfl(p1, a - 64, 64);
        float olp = old_logp[src], ad = adv[src];
        float ratio = expf(logp_a - olp);
        float clipped = fminf(fmaxf(ratio, 1.f - clip), 1.f + clip);
        float surr1 = ratio * ad, surr2 = clipped * ad;
        // d(-min(surr1,surr2))/d logp : torch.min splits ties, and inside the clip range surr2 carries the other half
        float g_logp;
        bool inside = (ratio >= 1.f - clip) && (ratio <= 1.f + clip);
        if (inside) g_logp = -(ad * ratio);
        else if (surr1 < surr2) g_logp = -(ad * ratio);
        else if (surr1 == surr2) g_logp = -(ad * ratio) * 0.5f;
        else g_logp = 0.f;
        // gradient wrt clamped probs p_i:  logp_a -> 1/p_a ; -ent_coef * H -> ent_coef * (log p_i + 1)
        float g0 = (lane < A) ? ent_coef * (lp0 + 1.f) : 0.f, g1 = ((lane + 64) < A) ? ent_coef * (lp1 + 1.f) : 0.f;
        if (lane == a) g0 += g_logp / p_a;
        if (lane + 64 == a) g1 += g_logp / p_a;
        // clamp backward: passes where 1e-11 <= s <= 1
        if (!(s0 >= 1e-11f && s0 <= 1.f)) g0 = 0.f;
        if (!(s1 >= 1e-11f && s1 <= 1.f)) g1 = 0.f;
        float dotgs = wave_sum(g0 * s0 + g1 * s1);
        float dz0 = s0 * (g0 - dotgs) * inv_temp * scale, dz1 = s1 * (g1 - dotgs) * inv_temp * scale;
        if (d16) {
            if (lane < ld16) d16[(size_t)row * ld16 + lane] = lane < A ? f2bf(dz0) : (short)0;
            if (lane + 64 < ld16) d16[(size_t)row * ld16 + lane + 64] = (lane + 64) < A ? f2bf(dz1) : (short)0;
        } else {
            if (lane < A) dlogits[(size_t)row * ld + lane] = dz0;
            if (lane + 64 < A) dlogits[(size_t)row * ld + lane + 64] = dz1;
        }
        float lr = logp_a - olp;
        m_ent += ent; m_kl += (expf(lr) - 1.f) - lr; m_clip += fabsf(ratio - 1.f) > clip ? 1.f : 0.f; m_ratio += ratio;
    }
    if (!metrics) return;
    if (lane == 0) { red[wave][0] = m_ent; red[wave][1] = m_kl; red[wave][2] = m_clip; red[wave][3] = m_ratio; }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&metrics[threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// value loss gradient: dL/dv = 2 (v - target) * scale ; metric[4] += (v-target)^2.  dv as fp32 [rows] or bf16 [rows][ld16] (column 0)
__global__ void __launch_bounds__(256) k_value_loss(const float* v, const float* targets, const int32_t* idx, int rows, float scale, float* dv, short* d16, int ld16, float* metrics) {
    __shared__ float red[4];
    float sq = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
        int src = idx ? idx[i] : i;
        float d = v[i] - targets[src];
        float gr = 2.f * d * scale;
        if (d16) { for (int c = 0; c < ld16; c++) d16[(size_t)i * ld16 + c] = c == 0 ? f2bf(gr) : (short)0; }
        else dv[i] = gr;
        sq += d * d;
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (metrics && threadIdx.x == 0) atomicAdd(&metrics[4], red[0] + red[1] + red[2] + red[3]);
}

__global__ void k_gather_rows(const float* src, const int32_t* idx, int rows, int D, float* dst) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * D) return;
    int r = (int)(i / D), c = (int)(i % D);
    dst[i] = src[(size_t)idx[r] * D + c];
}

// sum of squares of a gradient segment -> out[0] (fp32 atomics)
// The squared norm of the (scaled) gradient in a FIXED order: a partial per workgroup (wave sums, then the workgroup's waves in order), then the
// partials in order by one thread.  With atomics the order -- and with it the last bit of the clip coefficient -- changed from launch to
// launch: the replicas of a multi-GPU run, which start every optimizer step from bit-identical summed gradients, drifted apart by an ulp at a
// time (found by the replica check, tests/test_host_cpp.py::test_two_ranks_on_one_gpu_...).
constexpr int SUMSQ_BLOCKS = 64;
// The fp16 mode's dynamic loss scale and what hangs on it (GradScaler: gradscaler.hpp:26-34,162,291), resident on the device: an optimizer step
// decides there whether it happens (k_ls_decide), so rlgpu_clip_adam_step returns without waiting for the minibatch -- under
// collectionDuringLearn the host thread has the next collection to launch (with the decision on the host, fp16 + overlap measured no overlap at all).
struct LsDev {
    float scale; int growth; int skipped;      // the scale, clean steps since it last changed, steps skipped so far
    int skip[2]; float unscale;                // this step: skip optimizer 0 / 1; the scale the step's gradients carry
    float bc1[2], bc2s[2]; long long step[2];  // Adam's bias corrections for this step and the step counts (policy, critic)
};
__global__ void __launch_bounds__(256) k_sumsq(const float* g, int64_t n, float pre_scale, float* partial, const LsDev* ls) {
    __shared__ float ws[4];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ls) pre_scale /= ls->scale;
    float s = 0.f;
    for (; i < n; i += (int64_t)gridDim.x * blockDim.x) { float v = g[i] * pre_scale; s += v * v; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((ws[0] + ws[1]) + ws[2]) + ws[3];
}
__global__ void k_sumsq_finish(const float* partial, int n_partial, float* out) {
    float s = 0.f;
    for (int i = 0; i < n_partial; i++) s += partial[i];
    *out = s;
}
// clip_grad_norm_ (torch/nn/utils/clip_grad.py: coef = max_norm / (norm + 1e-6), clamped to 1) + Adam
// one thread, after both gradient norms are known: skip or step, back the scale off or count towards its next doubling
__global__ void k_ls_decide(const float* sumsq2, LsDev* ls, float b1, float b2) {
    const bool s0 = !isfinite(sumsq2[0]), s1 = !isfinite(sumsq2[1]);
    ls->skip[0] = s0; ls->skip[1] = s1; ls->unscale = ls->scale;
    if (s0 || s1) { ls->scale *= 0.5f; ls->growth = 0; ls->skipped++; }
    else if (++ls->growth >= 2000) { ls->scale *= 2.f; ls->growth = 0; }
    for (int k = 0; k < 2; k++) if (!ls->skip[k]) {
        const double t = (double)(++ls->step[k]);
        ls->bc1[k] = (float)(1.0 - pow((double)b1, t)); ls->bc2s[k] = (float)sqrt(1.0 - pow((double)b2, t));
    }
}
__global__ void k_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float pre_scale, float max_norm,
                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, const LsDev* ls, int slot) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (ls) { if (ls->skip[slot]) return; pre_scale /= ls->unscale; bc1 = ls->bc1[slot]; bc2_sqrt = ls->bc2s[slot]; }
    float norm = sqrtf(*sumsq);
    float coef = fminf(max_norm / (norm + 1e-6f), 1.f);
    float gr = g[i] * pre_scale * coef;
    float mi = b1 * m[i] + (1.f - b1) * gr;
    float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
}

struct Net {
    int n_layers = 0;              // number of Linear layers
    int dims[10] = {0};            // dims[0] = in, dims[n_layers] = out
    int64_t w_off[9] = {0}, b_off[9] = {0};
    int64_t n_params = 0;
    // bf16 fast path: padded leading dims and the offsets (in elements) of the weight shadows inside rlgpu_learner::shadows
    int kp[10] = {0};              // kp[i] = round_up(dims[i], 32): leading dim of layer i's bf16 input / of its gradient
    int64_t w16_off[9] = {0}, wt16_off[9] = {0}, wf16_off[9] = {0}, wtf16_off[9] = {0};   // wtf16: wt16 in fragment order too (mlp_stripe.h k_bwd_stripe)
    // (wf16: w16 re-ordered into MFMA B fragments: k_mlp_infer, k_fwd_stripe)
    int w16_rows[9] = {0}, wt16_rows[9] = {0};   // row counts padded to the 128-wide N tile
};

}  // namespace

struct rlgpu_learner {
    int device = 0;
    RlgpuLearnerConfig cfg{};
    Net pol, cri;
    int64_t n_total = 0;
    float *params = nullptr, *grads = nullptr, *adam_m = nullptr, *adam_v = nullptr;
    int64_t step_p = 0, step_c = 0;
    // scratch: activations per net [max_rows x width]
    std::vector<float*> act_p, act_c;  // act[i] = output of layer i (post-ReLU for hidden), act_p.back() = logits
    float *dbuf0 = nullptr, *dbuf1 = nullptr, *gathered = nullptr, *norm_buf = nullptr;
    // deterministic-gradient mode (rlgpu_learner_set_deterministic): every dW / db element has ONE writer per launch, or goes through per-slab partials
    // that are summed in slab order (dw_partial: [slabs][n_total], allocated on first use)
    bool deterministic = false; float* dw_partial = nullptr; size_t dw_partial_slabs = 0;
    // bf16 fast path (cfg.use_bf16)
    short* shadows = nullptr; int64_t n_shadow = 0; bool shadows_dirty = true;
    // fp16 operand mode (cfg.use_bf16 == 2; BASELINE configs[4] "fp16 autocast"): the minibatch kernels of ppo_fused.h take fp16 copies of the
    // weights (same layout as `shadows`) and the loss gradient times a dynamic loss scale (PRIV/Util/gradscaler.hpp:26-34: 2^16 at the start,
    // x 2 after 2000 steps without an overflow, x 0.5 and the step SKIPPED after one).  Inference (collection, value pass) stays bf16.
    short* shadows_h = nullptr; float loss_scale = 65536.f; int ls_growth = 0; int ls_skipped = 0;   // fp16 mode: host mirror of *ls_dev (ls_pull / ls_push)
    LsDev* ls_dev = nullptr;
    short* x16 = nullptr;                       // [max_rows][kp[0]] network input
    std::vector<short*> act16_p, act16_c;       // hidden activations [max_rows][kp[i+1]]
    short *g16a = nullptr, *g16b = nullptr;     // activation gradients, ping-pong [max_rows][max kp]
    // bf16 fast path, rlgpu_ppo_minibatch: the critic's chain runs beside the policy's on a side stream, and each network's dW GEMMs run
    // beside its dX chain on a stream of their own -- so every layer's activation gradient keeps its own buffer until the dW GEMM read it
    short* dy16[2][9] = {};                     // [policy, critic][layer]: dL/d(layer output), [max_rows][max kp]
    hipStream_t side = nullptr, dw_stream[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_dy[2][9] = {}, ev_dw_done[2] = {nullptr, nullptr};
    uint32_t call_ctr = 0, sampler_stream = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false; double last_flops = 0;
    bool timing_on = false;   // rlgpu_learner_enable_timing
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; size_t ev_used = 0; std::vector<double> ev_flops;
    double acc_ms = 0, acc_flops = 0; int acc_calls = 0;
    std::string err;
    struct Redzone { void* base; size_t bytes; std::string name; };
    std::vector<Redzone> redzones;   // RLGPU_REDZONE (debug): the guarded tail of every device buffer the learner owns (rlgpu_learner_check_redzones)
    size_t redzone_bytes = 0;
};

struct rlgpu_shuffler { std::default_random_engine rng; std::vector<int64_t> scratch; };

// ExperienceBuffer bookkeeping (PRIV/PPO/ExperienceBuffer.cpp:17-68): which rows of which submitted iteration are still in the FIFO
struct rlgpu_expbuf {
    int64_t max_rows = 0; int T = 0, n = 0; int64_t B = 0;   // B = rows of a device slot (T x n); T = steps of a lockstep iteration
    // rows [skip, rows) of the iteration stored in device slot `slot`, in agent-major order; off (n + 1 entries) = where each trajectory starts when
    // the trajectories have their own lengths (rlgpu_expbuf_submit_ragged), empty = n trajectories of T rows
    struct Chunk { int slot; int64_t skip; int64_t rows; std::vector<int64_t> off; };
    std::vector<Chunk> chunks;                  // oldest first
    int n_slots = 0;
};

#define LCHK(l, call)                                                                           \
    do {                                                                                         \
        hipError_t _s = (call);                                                                  \
        if (_s != hipSuccess) {                                                                  \
            (l)->err = std::string(#call) + ": " + hipGetErrorString(_s);                        \
            return RLGPU_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

namespace {

void build_net(Net& n, int in, const int32_t* hidden, int n_hidden, int out, int64_t& off) {
    n.n_layers = n_hidden + 1;
    n.dims[0] = in;
    for (int i = 0; i < n_hidden; i++) n.dims[i + 1] = hidden[i];
    n.dims[n.n_layers] = out;
    n.n_params = 0;
    for (int i = 0; i < n.n_layers; i++) {
        n.w_off[i] = off; off += (int64_t)n.dims[i] * n.dims[i + 1];
        n.b_off[i] = off; off += n.dims[i + 1];
        n.n_params += (int64_t)n.dims[i] * n.dims[i + 1] + n.dims[i + 1];
    }
}

int launch_gemm(rlgpu_learner* l, const GemmArgs& g, int splits) {
    dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits), block(GEMM_THREADS);
    if (l->cfg.use_bf16) hipLaunchKernelGGL(k_gemm<true>, grid, block, 0, l->stream, g);
    else hipLaunchKernelGGL(k_gemm<false>, grid, block, 0, l->stream, g);
    LCHK(l, hipGetLastError());
    l->last_flops += 2.0 * g.M * g.N * g.K;
    return RLGPU_OK;
}

// forward through `net`; acts[i] receives layer i's output (ld = dims[i+1])
int net_forward(rlgpu_learner* l, const Net& net, const std::vector<float*>& acts, const float* x, int rows) {
    const float* in = x; int ld_in = net.dims[0];
    for (int i = 0; i < net.n_layers; i++) {
        GemmArgs g{};
        g.A = in; g.lda = ld_in; g.a_trans = 0;
        g.B = l->params + net.w_off[i]; g.ldb = net.dims[i]; g.b_trans = 0;
        g.C = acts[i]; g.ldc = net.dims[i + 1];
        g.M = rows; g.N = net.dims[i + 1]; g.K = net.dims[i];
        g.bias = l->params + net.b_off[i]; g.relu = (i < net.n_layers - 1) ? 1 : 0;
        g.mask = nullptr; g.atomic_accumulate = 0; g.k_chunk = g.K;
        int rc = launch_gemm(l, g, 1);
        if (rc) return rc;
        in = acts[i]; ld_in = net.dims[i + 1];
    }
    return RLGPU_OK;
}

// backward: dout = dL/d(output of last layer) [rows x out]; accumulates into l->grads. Uses dbuf0/dbuf1 ping-pong.
int net_backward(rlgpu_learner* l, const Net& net, const std::vector<float*>& acts, const float* x, int rows, float* dout) {
    float* cur = dout;
    for (int i = net.n_layers - 1; i >= 0; i--) {
        const float* in = (i == 0) ? x : acts[i - 1];
        int K_in = net.dims[i], N_out = net.dims[i + 1];
        // dW[N_out x K_in] += cur^T[N_out x rows] . in[rows x K_in]   (split over rows, fp32 atomics)
        {
            GemmArgs g{};
            g.A = cur; g.lda = N_out; g.a_trans = 1;
            g.B = in; g.ldb = K_in; g.b_trans = 1;
            g.C = l->grads + net.w_off[i]; g.ldc = K_in;
            g.M = N_out; g.N = K_in; g.K = rows;
            g.atomic_accumulate = 1;
            int chunk = l->deterministic ? rows : 2048; g.k_chunk = chunk;   // (deterministic mode: one block per dW tile sums all rows in order; its single atomic add lands on a value nothing else touches meanwhile)
            int splits = (rows + chunk - 1) / chunk;
            int rc = launch_gemm(l, g, splits);
            if (rc) return rc;
        }
        {
            dim3 grid((N_out + 63) / 64, l->deterministic ? 1 : std::max(1, std::min(1024, rows / 64))), block(256);
            hipLaunchKernelGGL(k_col_sum, grid, block, 0, l->stream, (const float*)cur, N_out, rows, N_out, l->grads + net.b_off[i]);
            LCHK(l, hipGetLastError());
        }
        if (i > 0) {
            // dX[rows x K_in] = cur[rows x N_out] . W[N_out x K_in], masked by ReLU of acts[i-1]
            float* nxt = (cur == l->dbuf0) ? l->dbuf1 : l->dbuf0;
            GemmArgs g{};
            g.A = cur; g.lda = N_out; g.a_trans = 0;
            g.B = l->params + net.w_off[i]; g.ldb = K_in; g.b_trans = 1;
            g.C = nxt; g.ldc = K_in;
            g.M = rows; g.N = K_in; g.K = N_out;
            g.mask = acts[i - 1]; g.ldmask = K_in;
            g.k_chunk = g.K;
            int rc = launch_gemm(l, g, 1);
            if (rc) return rc;
            cur = nxt;
        }
    }
    return RLGPU_OK;
}

// ---- bf16 fast path, host side ------------------------------------------------------------------------------------
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

void plan_shadows(Net& n, int64_t& off) {
    for (int i = 0; i <= n.n_layers; i++) n.kp[i] = round_up(n.dims[i], 32);
    for (int i = 0; i < n.n_layers; i++) {
        n.w16_rows[i] = round_up(n.dims[i + 1], 128);   // B operand of the forward GEMM: [N padded][kp[i]]
        n.w16_off[i] = off; off += (int64_t)n.w16_rows[i] * n.kp[i];
        n.wt16_rows[i] = round_up(n.dims[i], 128);      // B operand of the dX GEMM: W^T as [K_in padded][kp[i+1]]
        n.wt16_off[i] = off; off += (int64_t)n.wt16_rows[i] * n.kp[i + 1];
        n.wf16_off[i] = off; off += (int64_t)n.w16_rows[i] * n.kp[i];
        n.wtf16_off[i] = off; off += (int64_t)n.wt16_rows[i] * n.kp[i + 1];
    }
}

int refresh_shadows(rlgpu_learner* l) {
    if (!l->shadows_dirty) return RLGPU_OK;
    for (int pass = 0; pass < (l->shadows_h ? 2 : 1); pass++) {      // bf16 copies; in fp16 mode the same four copies in fp16 for the minibatch kernels
        short* const base = pass == 0 ? l->shadows : l->shadows_h;
        RefreshArgs a{}; int nl = 0; size_t most = 0;
        for (const Net* n : {&l->pol, &l->cri})
            for (int i = 0; i < n->n_layers; i++) {
                RefreshLayer& L = a.L[nl++];
                L.W = l->params + n->w_off[i]; L.N = n->dims[i + 1]; L.K = n->dims[i];
                L.w16 = base + n->w16_off[i]; L.wt16 = base + n->wt16_off[i]; L.wf = base + n->wf16_off[i]; L.wtf = base + n->wtf16_off[i];
                L.rows16 = n->w16_rows[i]; L.ld16 = n->kp[i]; L.rowst = n->wt16_rows[i]; L.ldt = n->kp[i + 1];
                most = std::max(most, std::max((size_t)L.rows16 * L.ld16, (size_t)L.rowst * L.ldt));
            }
        const dim3 grid((unsigned)((most + 255) / 256), (unsigned)nl);
        if (pass == 0) hipLaunchKernelGGL(k_refresh_all<false>, grid, dim3(256), 0, l->stream, a);
        else hipLaunchKernelGGL(k_refresh_all<true>, grid, dim3(256), 0, l->stream, a);
        LCHK(l, hipGetLastError());
    }
    l->shadows_dirty = false;
    return RLGPU_OK;
}

template <int EPI>
int launch_nt(rlgpu_learner* l, NtArgs g, int real_k) {
    const int n_cover = std::max(g.N, EPI == 1 ? g.N : g.ldc16);
    if ((g.N > 32 || EPI == 2) && (size_t)((n_cover + 127) / 128) * ((g.M + 127) / 128) < 256) {
        // inference-sized M (8192 agent rows): 128x128 tiles would occupy half of the 256 CUs -- 64x64 tiles, 4x the workgroups
        dim3 grid((n_cover + 63) / 64, (g.M + 63) / 64);
        hipLaunchKernelGGL((k_gemm_nt<2, 2, 1, 1, EPI>), grid, dim3(256), 0, l->stream, g);
    } else if (g.N > 32 || EPI == 2) {
        // (Tried in round 3 and dropped: ONE column of 128 x 256 tiles for the 256-wide layers, so that the activation operand is read once
        // instead of twice -- 0.521 -> 0.546 ms per minibatch at two workgroups per CU, 0.638 at one: the second read of A comes from L2 /
        // MALL anyway, and a workgroup's ~10 us are load latency and epilogue around ~1 us of MFMA work, hidden only by workgroups in flight.)
        dim3 grid((n_cover + 127) / 128, (g.M + 127) / 128);
        hipLaunchKernelGGL((k_gemm_nt<2, 2, 2, 2, EPI>), grid, dim3(256), 0, l->stream, g);
    } else {
        dim3 grid(1, (g.M + 127) / 128);
        hipLaunchKernelGGL((k_gemm_nt<4, 1, 1, 1, EPI>), grid, dim3(256), 0, l->stream, g);
    }
    LCHK(l, hipGetLastError());
    l->last_flops += 2.0 * g.M * g.N * real_k;
    return RLGPU_OK;
}

// x fp32 [rows][D] (rows optionally gathered through idx) -> l->x16
int stage_input16(rlgpu_learner* l, const float* x, const int32_t* idx, int rows) {
    const int D = l->cfg.obs_size, ld = l->pol.kp[0];
    size_t tot = (size_t)rows * ld;
    hipLaunchKernelGGL(k_rows_to_bf16, dim3((tot + 255) / 256), dim3(256), 0, l->stream, x, idx, rows, D, l->x16, ld);
    LCHK(l, hipGetLastError());
    return RLGPU_OK;
}

// forward through `net` from l->x16; hidden activations -> acts16[i] (bf16), last layer -> out32 (fp32, ld = out dim)
int net_forward16(rlgpu_learner* l, const Net& net, const std::vector<short*>& acts16, float* out32, int rows) {
    int rc = refresh_shadows(l);
    if (rc) return rc;
    const short* in = l->x16;
    for (int i = 0; i < net.n_layers; i++) {
        const bool last = (i == net.n_layers - 1);
        NtArgs g{};
        g.A = in; g.lda = net.kp[i];
        g.B = l->shadows + net.w16_off[i]; g.ldb = net.kp[i];
        g.M = rows; g.N = net.dims[i + 1]; g.K = net.kp[i];
        g.bias = l->params + net.b_off[i];
        if (last) { g.C32 = out32; g.ldc32 = net.dims[i + 1]; rc = launch_nt<1>(l, g, net.dims[i]); }
        else { g.C16 = acts16[i]; g.ldc16 = net.kp[i + 1]; rc = launch_nt<0>(l, g, net.dims[i]); in = acts16[i]; }
        if (rc) return rc;
    }
    return RLGPU_OK;
}

// backward: dout16 = dL/d(last layer output) as bf16 [rows][kp[L]] (zero padded); accumulates dW / db into l->grads.
// which < 0: everything on l->stream, gradients ping-pong between g16a / g16b.  which = 0 / 1 (policy / critic, bf16 minibatch path): the dX
// chain stays on l->stream and writes layer i's input gradient to dy16[which][i - 1]; the dW GEMM of layer i goes to dw_stream[which]
// behind the event that marks its dY ready -- dW work (45 % of the section) leaves the chain's critical path.
int net_backward16(rlgpu_learner* l, const Net& net, const std::vector<short*>& acts16, int rows, short* dout16, int which = -1) {
    short* cur = dout16;
    const bool split = which >= 0 && l->dw_stream[which] != nullptr;
    for (int i = net.n_layers - 1; i >= 0; i--) {
        const short* in = (i == 0) ? l->x16 : acts16[i - 1];
        const int K_in = net.dims[i], N_out = net.dims[i + 1];
        hipStream_t dws = l->stream;
        if (split) {
            LCHK(l, hipEventRecord(l->ev_dy[which][i], l->stream));            // `cur` (this layer's dY) is complete on the chain
            LCHK(l, hipStreamWaitEvent(l->dw_stream[which], l->ev_dy[which][i], 0));
            dws = l->dw_stream[which];
        }
        {
            TnArgs t{};
            t.Y = cur; t.ldy = net.kp[i + 1]; t.X = in; t.ldx = net.kp[i];
            t.R = rows; t.Mo = N_out; t.No = K_in;
            t.dW = l->grads + net.w_off[i]; t.ldw = K_in; t.db = l->grads + net.b_off[i];
            // 512-row slabs: measured optimum between atomic traffic (128 rows: 99 TFLOP/s for the whole minibatch, 256: 139) and
            // too few workgroups (2048: 143); slab partials + a reduction kernel instead of atomics were slower (153 vs 171)
            t.slab = l->deterministic ? std::max(rows, 1) : 512;
            dim3 grid((K_in + 127) / 128, (N_out + 127) / 128, (rows + t.slab - 1) / t.slab);
            hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, dws, t);
            LCHK(l, hipGetLastError());
            l->last_flops += 2.0 * N_out * K_in * (double)rows;
        }
        if (i > 0) {
            short* nxt = split ? l->dy16[which][i - 1] : ((cur == l->g16a) ? l->g16b : l->g16a);
            NtArgs g{};
            g.A = cur; g.lda = net.kp[i + 1];
            g.B = l->shadows + net.wt16_off[i]; g.ldb = net.kp[i + 1];
            g.M = rows; g.N = K_in; g.K = net.kp[i + 1];
            g.C16 = nxt; g.ldc16 = net.kp[i];
            g.mask16 = acts16[i - 1]; g.ldm = net.kp[i];
            int rc = launch_nt<2>(l, g, N_out);
            if (rc) return rc;
            cur = nxt;
        }
    }
    if (split) LCHK(l, hipEventRecord(l->ev_dw_done[which], l->dw_stream[which]));
    return RLGPU_OK;
}

// ---- the stripe path (mlp_stripe.h): forward and dX chains of BOTH networks in one launch each -----------------------------------------------
#ifndef RLG_STRIPE_RT
#define RLG_STRIPE_RT 2          // 32-row tiles per stripe: 2 -> 64 rows, 68 KB of LDS, two workgroups per CU
#endif
bool stripe_capable(const rlgpu_learner* l) {
    // opt-in for now (RLGPU_STRIPE=1): the two fused launches give the per-layer kernels' results bit for bit, but with the dW GEMMs still
    // separate a minibatch takes 0.74 ms against the four-stream per-layer path's 0.52 (what is missing is dW inside the stripe)
    static const bool on = std::getenv("RLGPU_STRIPE") != nullptr;
    if (!on || !l->cfg.use_bf16) return false;
    for (const Net* n : {&l->pol, &l->cri}) {
        if (n->n_layers < 2 || n->n_layers > stripe::MAXL) return false;
        for (int i = 0; i <= n->n_layers; i++) if (n->kp[i] > stripe::MAXW || !stripe::nk_supported(n->kp[i] / 16)) return false;
    }
    return l->pol.kp[0] == l->cri.kp[0];
}
void stripe_fill(const rlgpu_learner* l, stripe::Args& a, int rows) {
    a.x16 = l->x16; a.ldx = l->pol.kp[0]; a.rows = rows;
    int w = 0;
    for (const Net* n : {&l->pol, &l->cri}) {
        stripe::Net& s = a.net[w];
        s.n_layers = n->n_layers;
        for (int i = 0; i <= n->n_layers; i++) { s.dims[i] = n->dims[i]; s.kp[i] = n->kp[i]; }
        const std::vector<short*>& acts = w == 0 ? l->act16_p : l->act16_c;
        for (int i = 0; i < n->n_layers; i++) {
            s.wf[i] = l->shadows + n->wf16_off[i]; s.wtf[i] = l->shadows + n->wtf16_off[i]; s.bias[i] = l->params + n->b_off[i];
            s.act[i] = i + 1 < n->n_layers ? acts[i] : nullptr;
            s.dy[i] = l->dy16[w][i];
        }
        s.out32 = w == 0 ? l->act_p.back() : l->act_c.back(); s.ld32 = n->dims[n->n_layers];
        w++;
    }
}
int stripe_launch(rlgpu_learner* l, bool backward, int rows) {
    constexpr int RT = RLG_STRIPE_RT;
    constexpr size_t smem = (size_t)2 * 32 * RT * stripe::LD * sizeof(short);
    static bool attr_set = false;
    if (!attr_set) {
        LCHK(l, hipFuncSetAttribute(reinterpret_cast<const void*>(&stripe::k_fwd_stripe<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        LCHK(l, hipFuncSetAttribute(reinterpret_cast<const void*>(&stripe::k_bwd_stripe<RT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    stripe::Args a; stripe_fill(l, a, rows);
    dim3 grid((rows + 32 * RT - 1) / (32 * RT), 2);
    if (backward) hipLaunchKernelGGL((stripe::k_bwd_stripe<RT>), grid, dim3(256), smem, l->stream, a);
    else hipLaunchKernelGGL((stripe::k_fwd_stripe<RT>), grid, dim3(256), smem, l->stream, a);
    LCHK(l, hipGetLastError());
    for (const Net* n : {&l->pol, &l->cri})
        for (int i = backward ? 1 : 0; i < n->n_layers; i++) l->last_flops += 2.0 * rows * (double)n->dims[i] * n->dims[i + 1];
    return RLGPU_OK;
}
// ---- the fused path (ppo_fused.h): gather + forward + loss + dX chain of both networks in ONE launch, every dW / db in a second one -------------
bool fused_capable(const rlgpu_learner* l) {
    static const bool off = std::getenv("RLGPU_NO_FUSED") != nullptr;
    if (off || !l->cfg.use_bf16) return false;
    for (const Net* n : {&l->pol, &l->cri}) {
        if (n->n_layers != 4) return false;
        for (int i = 1; i <= 3; i++) if (n->dims[i] != fused::H) return false;
    }
    const int k0 = l->pol.kp[0];
    if (k0 != l->cri.kp[0] || (k0 != 96 && k0 != 128 && k0 != 192)) return false;   // 1v1 / 2v2-padded / 3v3-padded observation rows
    return l->pol.kp[4] == 96 && l->cri.dims[4] == 1;
}
template <int K0P, bool HALF>
int fused_launch_th(rlgpu_learner* l, const fused::Args& a, dim3 grid) {
    static bool attr_set = false;
    if (!attr_set) {
        LCHK(l, hipFuncSetAttribute(reinterpret_cast<const void*>(&fused::k_ppo_fwd_bwd<K0P, 96, HALF>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused::SMEM_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((fused::k_ppo_fwd_bwd<K0P, 96, HALF>), grid, dim3(512), fused::SMEM_BYTES, l->stream, a);
    LCHK(l, hipGetLastError());
    return RLGPU_OK;
}
int fused_minibatch(rlgpu_learner* l, const float* obs, const int32_t* actions, const float* old_logp, const float* adv, const float* targets,
                    const int32_t* idx, int n, float ratio, float* metrics) {
    int rc = refresh_shadows(l);
    if (rc) return rc;
    fused::Args a{};
    a.obs = obs; a.idx = idx; a.rows = n; a.D = l->cfg.obs_size; a.x16 = l->x16;
    a.actions = actions; a.old_logp = old_logp; a.adv = adv; a.targets = targets;
    a.inv_temp = 1.0f / (l->cfg.temperature > 0 ? l->cfg.temperature : 1.f); a.clip = l->cfg.clip_range; a.ent_coef = l->cfg.ent_coef;
    a.scale = ratio / (float)n; a.metrics = metrics; a.loss_scale = 1.f; a.loss_scale_dev = l->ls_dev ? &l->ls_dev->scale : nullptr;
    static const int fz_debug = RLGPU_EXPERIMENT_ENV("RLGPU_FZ_DEBUG") ? std::atoi(RLGPU_EXPERIMENT_ENV("RLGPU_FZ_DEBUG")) : 0;
    a.debug = fz_debug;
    fused::DwArgs d{};
    d.rows = n;
    static const int slab_env = RLGPU_EXPERIMENT_ENV("RLGPU_DW_SLAB") ? std::atoi(RLGPU_EXPERIMENT_ENV("RLGPU_DW_SLAB")) : 0;
    d.slab = slab_env > 0 ? (slab_env + 31) / 32 * 32 : 2048;   // measured: the fp32 atomics of a flush cost 173 us per minibatch at 512 rows, 33 at 2048
    // at most 64 slabs, i.e. 64 partial gradient buffers (1.33 MB each at 256 x 3): a minibatch of millions of rows (BASELINE configs[4]) would otherwise ask
    // for gigabytes of partials, re-read by the reduction; the slab grows instead (whole 32-row tiles)
    constexpr int DW_MAX_SLABS = 64;
    if ((n + d.slab - 1) / d.slab > DW_MAX_SLABS) d.slab = ((n + DW_MAX_SLABS - 1) / DW_MAX_SLABS + 31) / 32 * 32;
    static const int dw_debug = RLGPU_EXPERIMENT_ENV("RLGPU_DW_DEBUG") ? std::atoi(RLGPU_EXPERIMENT_ENV("RLGPU_DW_DEBUG")) : 0;
    d.debug = dw_debug;
    int w = 0;
    for (const Net* nn : {&l->pol, &l->cri}) {
        fused::NetArgs& s = a.net[w];
        const std::vector<short*>& acts = w == 0 ? l->act16_p : l->act16_c;
        for (int i = 0; i < 4; i++) {
            const short* const sh = l->shadows_h ? l->shadows_h : l->shadows;
            s.wf[i] = sh + nn->wf16_off[i]; s.wtf[i] = sh + nn->wtf16_off[i]; s.bias[i] = l->params + nn->b_off[i];
            if (i < 3) s.act[i] = acts[i];
            s.dy[i] = l->dy16[w][i];
            fused::DwLayer& L = d.L[w][i];
            L.Y = l->dy16[w][i]; L.ldy = nn->kp[i + 1]; L.X = i == 0 ? l->x16 : acts[i - 1]; L.ldx = nn->kp[i];
            L.Mo = nn->dims[i + 1]; L.No = nn->dims[i]; L.dW = l->grads + nn->w_off[i]; L.db = l->grads + nn->b_off[i];
            l->last_flops += (i > 0 ? 6.0 : 4.0) * n * (double)nn->dims[i] * nn->dims[i + 1];   // forward + dW (+ dX above layer 0)
        }
        s.out_dim = nn->dims[4];
        w++;
    }
    // The two kernels alternate over chunks of rows (RLGPU_FUSED_CHUNK, 0 = the whole minibatch at once): what the first writes for a chunk is
    // still in the last-level cache when the second reads it
    static const int chunk_env = RLGPU_EXPERIMENT_ENV("RLGPU_FUSED_CHUNK") ? std::atoi(RLGPU_EXPERIMENT_ENV("RLGPU_FUSED_CHUNK")) : 0;
    const int chunk = chunk_env > 0 ? (chunk_env + fused::R - 1) / fused::R * fused::R : n;
    static const bool prof_on = RLGPU_EXPERIMENT_ENV("RLGPU_FUSED_PROF") != nullptr;   // (tools only, with a -DFZ_PROF build: per-phase cycles of k_ppo_fwd_bwd, printed every 16 calls)
    static unsigned long long* prof_buf = nullptr; static int prof_calls = 0;
    if (prof_on) {
        if (!prof_buf) { LCHK(l, hipMalloc(&prof_buf, 32 * 8)); LCHK(l, hipMemset(prof_buf, 0, 32 * 8)); }
        a.prof = prof_buf;
    }
    static bool dw_attr = false;
    if (!dw_attr) {
        LCHK(l, hipFuncSetAttribute(reinterpret_cast<const void*>(&fused::k_dw_grouped<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused::DW_SMEM_BYTES));
        LCHK(l, hipFuncSetAttribute(reinterpret_cast<const void*>(&fused::k_dw_grouped<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused::DW_SMEM_BYTES));
        dw_attr = true;
    }
    const dim3 grid_all((n + fused::R - 1) / fused::R, 2);
    for (int r0 = 0; r0 < n; r0 += chunk) {
        const int r1 = std::min(n, r0 + chunk);
        a.row0 = r0; a.rows = r1; d.row0 = r0; d.rows = r1;
        const dim3 grid((r1 - r0 + fused::R - 1) / fused::R, 2);
        const bool half = l->shadows_h != nullptr;
        switch (l->pol.kp[0]) {
            case 96: rc = half ? fused_launch_th<96, true>(l, a, grid) : fused_launch_th<96, false>(l, a, grid); break;
            case 128: rc = half ? fused_launch_th<128, true>(l, a, grid) : fused_launch_th<128, false>(l, a, grid); break;
            default: rc = half ? fused_launch_th<192, true>(l, a, grid) : fused_launch_th<192, false>(l, a, grid); break;
        }
        if (rc) return rc;
        const int n_slabs = (r1 - r0 + d.slab - 1) / d.slab;
        d.partial = nullptr; d.partial_stride = (size_t)l->n_total; d.grads_base = l->grads;
        // Per-slab partial gradients + a fixed-order sum are this path's ONLY form since round 5: measured at the flagship shape they cost no more than
        // the fp32 atomics they replace (ppo_iter_ms 1.63 vs 1.69, one box, alternating) and the gradient no longer depends on the order in which
        // the slabs finish.  (A make EXPERIMENTS=1 build can go back to atomics with RLGPU_DW_ATOMICS=1 for A/B runs.)
        static const bool dw_atomics = RLGPU_EXPERIMENT_ENV("RLGPU_DW_ATOMICS") != nullptr;
        if (!dw_atomics || l->deterministic) {
            if (l->dw_partial_slabs < (size_t)n_slabs) {
                if (l->dw_partial) (void)hipFree(l->dw_partial);
                l->dw_partial = nullptr; l->dw_partial_slabs = 0;
                LCHK(l, hipMalloc(&l->dw_partial, (size_t)n_slabs * (size_t)l->n_total * sizeof(float)));
                l->dw_partial_slabs = (size_t)n_slabs;
            }
            d.partial = l->dw_partial;
        }
        if (half) hipLaunchKernelGGL(fused::k_dw_grouped<true>, dim3(n_slabs, 4, 2), dim3(512), fused::DW_SMEM_BYTES, l->stream, d);
        else hipLaunchKernelGGL(fused::k_dw_grouped<false>, dim3(n_slabs, 4, 2), dim3(512), fused::DW_SMEM_BYTES, l->stream, d);
        LCHK(l, hipGetLastError());
        if (d.partial) {   // (every element of both networks' gradients was stored by every slab: the fused path covers all eight layers)
            hipLaunchKernelGGL(fused::k_dw_reduce, dim3((unsigned)((l->n_total + 255) / 256)), dim3(256), 0, l->stream, (const float*)d.partial, d.partial_stride, n_slabs, l->grads, (long long)l->n_total);
            LCHK(l, hipGetLastError());
        }
    }
    const dim3 grid = grid_all;
    if (prof_on && ++prof_calls % 16 == 0) {
        unsigned long long h[32];
        LCHK(l, hipDeviceSynchronize()); LCHK(l, hipMemcpy(h, prof_buf, sizeof(h), hipMemcpyDeviceToHost)); LCHK(l, hipMemset(prof_buf, 0, 32 * 8));
        const double wg = 16.0 * grid.x;
        static const char* names[10] = {"gather", "L0", "L1", "L2", "L3", "loss", "bwd3", "bwd2", "bwd1", "store dy0"};
        for (int net = 0; net < 2; net++) {
            fprintf(stderr, "k_ppo_fwd_bwd %s cycles/workgroup:", net ? "critic" : "policy");
            double tot = 0;
            for (int k = 0; k < 10; k++) { fprintf(stderr, " %s %.0f", names[k], h[net * 16 + k] / wg); tot += h[net * 16 + k] / wg; }
            fprintf(stderr, " | sum %.0f\n", tot);
        }
    }
    return RLGPU_OK;
}
// dW / db of every layer of one network from the stored activations and activation gradients (the TN GEMMs of net_backward16)
int net_dw16(rlgpu_learner* l, const Net& net, const std::vector<short*>& acts16, int rows, int which, hipStream_t st) {
    for (int i = net.n_layers - 1; i >= 0; i--) {
        const short* in = (i == 0) ? l->x16 : acts16[i - 1];
        const int K_in = net.dims[i], N_out = net.dims[i + 1];
        TnArgs t{};
        t.Y = l->dy16[which][i]; t.ldy = net.kp[i + 1]; t.X = in; t.ldx = net.kp[i];
        t.R = rows; t.Mo = N_out; t.No = K_in;
        t.dW = l->grads + net.w_off[i]; t.ldw = K_in; t.db = l->grads + net.b_off[i];
        t.slab = l->deterministic ? std::max(rows, 1) : 512;
        dim3 grid((K_in + 127) / 128, (N_out + 127) / 128, (rows + t.slab - 1) / t.slab);
        hipLaunchKernelGGL(k_gemm_tn, grid, dim3(256), 0, st, t);
        LCHK(l, hipGetLastError());
        l->last_flops += 2.0 * N_out * K_in * (double)rows;
    }
    return RLGPU_OK;
}

}  // namespace

extern "C" {

// Debug mode RLGPU_REDZONE=<bytes> (read at create), as for the env batch (rlgpu_env.hip): guard bytes behind every device buffer of the learner.
static int ls_push(rlgpu_learner* l);
static hipError_t lz_malloc(rlgpu_learner* l, void** p, size_t bytes, const char* name) {
    const size_t rz = l->redzone_bytes;
    hipError_t r = hipMalloc(p, bytes + rz);
    if (r != hipSuccess || !rz) return r;
    r = hipMemset((char*)*p + bytes, 0xC5, rz);
    l->redzones.push_back({*p, bytes, name});
    return r;
}
int rlgpu_learner_check_redzones(rlgpu_learner* l) {
    if (!l->redzone_bytes) { l->err = "rlgpu_learner_check_redzones: the learner was created without RLGPU_REDZONE"; return RLGPU_ERR_STATE; }
    LCHK(l, hipSetDevice(l->device));
    LCHK(l, hipDeviceSynchronize());
    std::vector<unsigned char> h(l->redzone_bytes);
    for (const auto& z : l->redzones) {
        LCHK(l, hipMemcpy(h.data(), (const char*)z.base + z.bytes, h.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++) if (h[i] != 0xC5) {
            size_t last = i; for (size_t j = i; j < h.size(); j++) if (h[j] != 0xC5) last = j;
            l->err = "redzone of '" + z.name + "' (" + std::to_string(z.bytes) + " bytes) overwritten: first at +" + std::to_string(i) + ", last at +" + std::to_string(last) + " past its end";
            return RLGPU_ERR_STATE;
        }
    }
    return RLGPU_OK;
}

int rlgpu_learner_create(rlgpu_learner** out, int device, const RlgpuLearnerConfig* cfg) {
    if (!out || !cfg || cfg->obs_size <= 0 || cfg->n_actions <= 0 || cfg->n_actions > 128 || cfg->max_rows <= 0) return RLGPU_ERR_ARG;
    if (cfg->n_policy_layers < 0 || cfg->n_policy_layers > 8 || cfg->n_critic_layers < 0 || cfg->n_critic_layers > 8) return RLGPU_ERR_ARG;
    rlgpu_learner* l = new rlgpu_learner();
    *out = l;
    l->device = device; l->cfg = *cfg;
    LCHK(l, hipSetDevice(device));
    { const char* rz = getenv("RLGPU_REDZONE"); l->redzone_bytes = rz ? (size_t)atol(rz) : 0; }
    int64_t off = 0;
    build_net(l->pol, cfg->obs_size, cfg->policy_layers, cfg->n_policy_layers, cfg->n_actions, off);
    build_net(l->cri, cfg->obs_size, cfg->critic_layers, cfg->n_critic_layers, 1, off);
    l->n_total = off;
    LCHK(l, lz_malloc(l, (void**)&l->params, (size_t)(off * 4), "params")); LCHK(l, lz_malloc(l, (void**)&l->grads, (size_t)(off * 4), "grads"));
    LCHK(l, lz_malloc(l, (void**)&l->adam_m, (size_t)(off * 4), "adam_m")); LCHK(l, lz_malloc(l, (void**)&l->adam_v, (size_t)(off * 4), "adam_v"));
    LCHK(l, hipMemset(l->grads, 0, off * 4)); LCHK(l, hipMemset(l->adam_m, 0, off * 4)); LCHK(l, hipMemset(l->adam_v, 0, off * 4));
    // torch nn::Linear default init: W, b ~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)) (kaiming_uniform(a=sqrt 5)); Philox stream
    std::vector<float> h(off);
    uint32_t ctr = 0;
    auto init_net = [&](const Net& n) {
        for (int i = 0; i < n.n_layers; i++) {
            float bound = 1.0f / std::sqrt((float)n.dims[i]);
            int64_t cnt = (int64_t)n.dims[i] * n.dims[i + 1] + n.dims[i + 1];
            for (int64_t k = 0; k < cnt; k += 4) {
                uint32_t r[4]; rlg::philox4(cfg->seed_lo, cfg->seed_hi, 0x1A17u, ctr++, 0, r);
                for (int q = 0; q < 4 && k + q < cnt; q++) h[n.w_off[i] + k + q] = (rlg::u32_to_unit(r[q]) * 2.f - 1.f) * bound;
            }
        }
    };
    init_net(l->pol); init_net(l->cri);
    LCHK(l, hipMemcpy(l->params, h.data(), off * 4, hipMemcpyHostToDevice));
    size_t R = (size_t)cfg->max_rows;
    int maxw = std::max(cfg->obs_size, cfg->n_actions);
    for (int i = 0; i < l->pol.n_layers; i++) { float* p; LCHK(l, lz_malloc(l, (void**)&p, (size_t)(R * l->pol.dims[i + 1] * 4), "policy activations")); l->act_p.push_back(p); maxw = std::max(maxw, l->pol.dims[i + 1]); }
    for (int i = 0; i < l->cri.n_layers; i++) { float* p; LCHK(l, lz_malloc(l, (void**)&p, (size_t)(R * l->cri.dims[i + 1] * 4), "critic activations")); l->act_c.push_back(p); maxw = std::max(maxw, l->cri.dims[i + 1]); }
    LCHK(l, lz_malloc(l, (void**)&l->dbuf0, (size_t)(R * maxw * 4), "dbuf0")); LCHK(l, lz_malloc(l, (void**)&l->dbuf1, (size_t)(R * maxw * 4), "dbuf1"));
    LCHK(l, lz_malloc(l, (void**)&l->gathered, (size_t)(R * cfg->obs_size * 4), "gathered"));
    LCHK(l, lz_malloc(l, (void**)&l->norm_buf, (size_t)((4 + 2 * SUMSQ_BLOCKS) * 4), "norm_buf"));   // [0..1] the two networks' squared gradient norms, then their per-workgroup partials
    if (cfg->use_bf16 < 0 || cfg->use_bf16 > 2) { l->err = "use_bf16: 0 (fp32), 1 (bf16) or 2 (fp16 operands + dynamic loss scale)"; return RLGPU_ERR_ARG; }
    if (cfg->use_bf16) {
        int64_t soff = 0;
        plan_shadows(l->pol, soff); plan_shadows(l->cri, soff);
        l->n_shadow = soff;
        LCHK(l, lz_malloc(l, (void**)&l->shadows, (size_t)(soff * 2), "shadows"));
        if (cfg->use_bf16 == 2) {
            // (kp is planned now: the shape test of fused_capable applies)
            if (!fused_capable(l)) { l->err = "use_bf16 = 2 (fp16 operands): only the shape the fused minibatch kernels cover (obs <= 192 padded, 256 x 3 hidden, n_actions 65..96)"; return RLGPU_ERR_ARG; }
            LCHK(l, lz_malloc(l, (void**)&l->shadows_h, (size_t)(soff * 2), "shadows_h"));
            LCHK(l, lz_malloc(l, (void**)&l->ls_dev, sizeof(LsDev), "loss scale state"));
            { int rc = ls_push(l); if (rc) return rc; }      // scale 65536, no steps counted, Adam's step counts 0
        }
        int maxkp = l->pol.kp[0];
        for (const Net* n : {&l->pol, &l->cri}) for (int i = 0; i <= n->n_layers; i++) maxkp = std::max(maxkp, n->kp[i]);
        LCHK(l, lz_malloc(l, (void**)&l->x16, (size_t)(R * l->pol.kp[0] * 2), "x16"));
        for (int i = 0; i + 1 < l->pol.n_layers; i++) { short* p; LCHK(l, lz_malloc(l, (void**)&p, (size_t)(R * l->pol.kp[i + 1] * 2), "policy activations (16-bit)")); l->act16_p.push_back(p); }
        for (int i = 0; i + 1 < l->cri.n_layers; i++) { short* p; LCHK(l, lz_malloc(l, (void**)&p, (size_t)(R * l->cri.kp[i + 1] * 2), "critic activations (16-bit)")); l->act16_c.push_back(p); }
        LCHK(l, lz_malloc(l, (void**)&l->g16a, (size_t)(R * maxkp * 2), "g16a")); LCHK(l, lz_malloc(l, (void**)&l->g16b, (size_t)(R * maxkp * 2), "g16b"));
        for (int w = 0; w < 2; w++) {
            const Net& net = w == 0 ? l->pol : l->cri;
            for (int i = 0; i < net.n_layers; i++) {
                LCHK(l, lz_malloc(l, (void**)&l->dy16[w][i], (size_t)(R * maxkp * 2), "dy16"));
                LCHK(l, hipEventCreateWithFlags(&l->ev_dy[w][i], hipEventDisableTiming));
            }
            LCHK(l, hipStreamCreateWithFlags(&l->dw_stream[w], hipStreamNonBlocking));
            LCHK(l, hipEventCreateWithFlags(&l->ev_dw_done[w], hipEventDisableTiming));
        }
        LCHK(l, hipStreamCreateWithFlags(&l->side, hipStreamNonBlocking));
        LCHK(l, hipEventCreateWithFlags(&l->ev_fork, hipEventDisableTiming)); LCHK(l, hipEventCreateWithFlags(&l->ev_join, hipEventDisableTiming));
        l->shadows_dirty = true;
    }
    return RLGPU_OK;
}

void rlgpu_learner_destroy(rlgpu_learner* l) {
    if (!l) return;
    (void)hipSetDevice(l->device);
    for (float* p : {l->params, l->grads, l->adam_m, l->adam_v, l->dbuf0, l->dbuf1, l->gathered, l->norm_buf, l->dw_partial}) if (p) (void)hipFree(p);
    for (float* p : l->act_p) (void)hipFree(p);
    for (float* p : l->act_c) (void)hipFree(p);
    for (short* p : {l->shadows, l->shadows_h, l->x16, l->g16a, l->g16b}) if (p) (void)hipFree(p);
    if (l->ls_dev) (void)hipFree(l->ls_dev);
    for (int w = 0; w < 2; w++) {
        for (int i = 0; i < 9; i++) { if (l->dy16[w][i]) (void)hipFree(l->dy16[w][i]); if (l->ev_dy[w][i]) (void)hipEventDestroy(l->ev_dy[w][i]); }
        if (l->dw_stream[w]) (void)hipStreamDestroy(l->dw_stream[w]);
        if (l->ev_dw_done[w]) (void)hipEventDestroy(l->ev_dw_done[w]);
    }
    if (l->side) (void)hipStreamDestroy(l->side);
    if (l->ev_fork) (void)hipEventDestroy(l->ev_fork);
    if (l->ev_join) (void)hipEventDestroy(l->ev_join);
    for (short* p : l->act16_p) (void)hipFree(p);
    for (short* p : l->act16_c) (void)hipFree(p);
    for (auto& p : l->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete l;
}
const char* rlgpu_learner_last_error(const rlgpu_learner* l) { return l ? l->err.c_str() : "null learner"; }
int rlgpu_learner_set_stream(rlgpu_learner* l, void* s) { l->stream = (hipStream_t)s; return RLGPU_OK; }
int64_t rlgpu_learner_num_params(const rlgpu_learner* l, int which) { return which == 0 ? l->pol.n_params : (which == 1 ? l->cri.n_params : l->n_total); }

static void seg(const rlgpu_learner* l, int which, int64_t& off, int64_t& n) {
    if (which == 0) { off = 0; n = l->pol.n_params; } else if (which == 1) { off = l->pol.n_params; n = l->cri.n_params; } else { off = 0; n = l->n_total; }
}
int rlgpu_learner_get_params(rlgpu_learner* l, int which, float* host) {
    int64_t off, n; seg(l, which, off, n); LCHK(l, hipSetDevice(l->device));
    LCHK(l, hipStreamSynchronize(l->stream));
    LCHK(l, hipMemcpy(host, l->params + off, n * 4, hipMemcpyDeviceToHost)); return RLGPU_OK;
}
int rlgpu_learner_set_params(rlgpu_learner* l, int which, const float* host) {
    int64_t off, n; seg(l, which, off, n); LCHK(l, hipSetDevice(l->device));
    LCHK(l, hipStreamSynchronize(l->stream));
    LCHK(l, hipMemcpy(l->params + off, host, n * 4, hipMemcpyHostToDevice)); l->shadows_dirty = true; return RLGPU_OK;
}
int rlgpu_learner_get_grads(rlgpu_learner* l, int which, float* host) {
    int64_t off, n; seg(l, which, off, n); LCHK(l, hipSetDevice(l->device));
    LCHK(l, hipStreamSynchronize(l->stream));
    LCHK(l, hipMemcpy(host, l->grads + off, n * 4, hipMemcpyDeviceToHost)); return RLGPU_OK;
}
int rlgpu_learner_grad_buffer(rlgpu_learner* l, float** p, int64_t* n) { *p = l->grads; *n = l->n_total; return RLGPU_OK; }
int rlgpu_learner_param_buffer(rlgpu_learner* l, float** p, int64_t* n) { *p = l->params; *n = l->n_total; l->shadows_dirty = true; /* the caller may write (broadcast) */ return RLGPU_OK; }
// fp16 mode: the loss scale, its counters and Adam's step counts live on the device (LsDev); the host fields are a mirror, refreshed on demand
static int ls_pull(rlgpu_learner* l) {
    if (!l->ls_dev) return RLGPU_OK;
    LsDev h;
    LCHK(l, hipMemcpyAsync(&h, l->ls_dev, sizeof(h), hipMemcpyDeviceToHost, l->stream)); LCHK(l, hipStreamSynchronize(l->stream));
    l->loss_scale = h.scale; l->ls_growth = h.growth; l->ls_skipped = h.skipped; l->step_p = h.step[0]; l->step_c = h.step[1];
    return RLGPU_OK;
}
static int ls_push(rlgpu_learner* l) {
    if (!l->ls_dev) return RLGPU_OK;
    LsDev h{}; h.scale = l->loss_scale; h.growth = l->ls_growth; h.skipped = l->ls_skipped; h.unscale = l->loss_scale; h.step[0] = l->step_p; h.step[1] = l->step_c;
    LCHK(l, hipMemcpyAsync(l->ls_dev, &h, sizeof(h), hipMemcpyHostToDevice, l->stream)); LCHK(l, hipStreamSynchronize(l->stream));
    return RLGPU_OK;
}
int rlgpu_learner_get_adam_state(rlgpu_learner* l, float* hm, float* hv, int64_t* sp, int64_t* sc) {
    LCHK(l, hipSetDevice(l->device)); LCHK(l, hipStreamSynchronize(l->stream));
    { int rc = ls_pull(l); if (rc) return rc; }
    if (hm) LCHK(l, hipMemcpy(hm, l->adam_m, l->n_total * 4, hipMemcpyDeviceToHost));
    if (hv) LCHK(l, hipMemcpy(hv, l->adam_v, l->n_total * 4, hipMemcpyDeviceToHost));
    if (sp) *sp = l->step_p; if (sc) *sc = l->step_c; return RLGPU_OK;
}
int rlgpu_learner_set_adam_state(rlgpu_learner* l, const float* hm, const float* hv, int64_t sp, int64_t sc) {
    LCHK(l, hipSetDevice(l->device)); LCHK(l, hipStreamSynchronize(l->stream));
    if (hm) LCHK(l, hipMemcpy(l->adam_m, hm, l->n_total * 4, hipMemcpyHostToDevice));
    if (hv) LCHK(l, hipMemcpy(l->adam_v, hv, l->n_total * 4, hipMemcpyHostToDevice));
    l->step_p = sp; l->step_c = sc; return ls_push(l);
}

// the fused inference kernel (k_mlp_infer) covers this net?  (bf16 path, LDS for two activation buffers, head width)
// the fused inference kernel (k_mlp_infer) covers this net?  (bf16 path, LDS for two activation buffers, head width)
static size_t fused_smem(const Net& net) {
    int maxkp = 0;
    for (int i = 0; i < net.n_layers; i++) maxkp = std::max(maxkp, net.kp[i]);
    return (size_t)std::max(FI_ROWS * (maxkp + 8), FI_ROWS * FI_LOGIT_LD * 2) * 2 * sizeof(short);
}
static bool fused_infer_fits(const rlgpu_learner* l, const Net& net, int out_dim) {
    if (!l->cfg.use_bf16 || getenv("RLGPU_NO_FUSED_INFER")) return false;
    return out_dim <= 128 && net.n_layers <= 9 && fused_smem(net) <= 65536;   // the default dynamic-LDS limit of a launch
}
static int launch_fused_infer(rlgpu_learner* l, const Net& net, const float* obs, int rows, int mode, float* values, const HeadArgs& head) {
    int rc = refresh_shadows(l);
    if (rc) return rc;
    FusedInferArgs g{};
    g.obs = obs; g.D = l->cfg.obs_size; g.rows = rows; g.n_layers = net.n_layers;
    int maxkp = 0;
    for (int i = 0; i < net.n_layers; i++) {
        g.W[i] = l->shadows + net.wf16_off[i]; g.bias[i] = l->params + net.b_off[i];
        g.K[i] = net.kp[i]; g.N[i] = net.dims[i + 1]; g.Npad[i] = net.kp[i + 1];
        maxkp = std::max(maxkp, net.kp[i]);
    }
    g.ld = maxkp + 8;
    g.buf_elems = std::max(FI_ROWS * g.ld, FI_ROWS * FI_LOGIT_LD * 2);
    g.mode = mode; g.values = values; g.head = head;
    g.stamps = (mode == 0 && RLGPU_EXPERIMENT_ENV("RLGPU_FUSED_STAMPS")) ? reinterpret_cast<unsigned long long*>(l->grads) : nullptr;   // debug: lands in the gradient buffer (tools/fused_infer_check.py)
    hipLaunchKernelGGL(k_mlp_infer, dim3((rows + FI_ROWS - 1) / FI_ROWS), dim3(64 * FI_WAVES), fused_smem(net), l->stream, g);
    LCHK(l, hipGetLastError());
    return RLGPU_OK;
}

static int policy_head(rlgpu_learner* l, const float* obs, int rows, int deterministic, const float* noise, int32_t* actions, float* logp, float* probs) {
    if (rows <= 0 || rows > l->cfg.max_rows) { l->err = "rows out of range (max_rows)"; return RLGPU_ERR_ARG; }
    LCHK(l, hipSetDevice(l->device));
    int rc;
    const int A = l->cfg.n_actions;
    const float inv_t = 1.0f / (l->cfg.temperature > 0 ? l->cfg.temperature : 1.f);
    HeadArgs h{A, inv_t, deterministic, noise, l->cfg.seed_lo, (l->cfg.seed_hi ^ 0x5A3C0DEu) + l->sampler_stream * 0x9E3779B9u, l->call_ctr, actions, logp, probs};
    l->call_ctr++;
    if (fused_infer_fits(l, l->pol, A)) return launch_fused_infer(l, l->pol, obs, rows, 0, nullptr, h);
    if (l->cfg.use_bf16) { if ((rc = stage_input16(l, obs, nullptr, rows))) return rc; rc = net_forward16(l, l->pol, l->act16_p, l->act_p.back(), rows); }
    else rc = net_forward(l, l->pol, l->act_p, obs, rows);
    if (rc) return rc;
    dim3 grid((rows + 7) / 8), block(256);   // 4 wavefronts x 2 rows
    hipLaunchKernelGGL(k_policy_head, grid, block, 0, l->stream, (const float*)l->act_p.back(), A, rows, h);
    LCHK(l, hipGetLastError());
    return RLGPU_OK;
}
// 1: policy and value inference run in the fused kernel (activations in its own LDS) -- such calls may go to another stream than a PPO
// epoch.  0: they use the learner's activation scratch (fp32 mode, nets too wide for the kernel), which rlgpu_ppo_minibatch uses too.
int rlgpu_learner_inference_is_standalone(const rlgpu_learner* l) {
    return (l && fused_infer_fits(l, l->pol, l->cfg.n_actions) && fused_infer_fits(l, l->cri, 1)) ? 1 : 0;
}
int rlgpu_policy_act(rlgpu_learner* l, const float* obs, int rows, int deterministic, const float* noise, int32_t* actions, float* logp) {
    if (!actions || !logp) return RLGPU_ERR_ARG;
    return policy_head(l, obs, rows, deterministic, noise, actions, logp, nullptr);
}
int rlgpu_policy_probs(rlgpu_learner* l, const float* obs, int rows, float* probs) { return policy_head(l, obs, rows, 1, nullptr, nullptr, nullptr, probs); }

int rlgpu_value_forward(rlgpu_learner* l, const float* obs, int rows, float* values) {
    if (rows <= 0 || rows > l->cfg.max_rows) { l->err = "rows out of range (max_rows)"; return RLGPU_ERR_ARG; }
    LCHK(l, hipSetDevice(l->device));
    int rc;
    // one fused launch; it also keeps this call off the activation scratch, so it may run on another stream than a PPO epoch
    // (collectionDuringLearn).  RLGPU_FUSED_VALUE_ROWS caps the row count that takes this path (experiments).
    static const int fused_cap = RLGPU_EXPERIMENT_ENV("RLGPU_FUSED_VALUE_ROWS") ? atoi(RLGPU_EXPERIMENT_ENV("RLGPU_FUSED_VALUE_ROWS")) : 0x7fffffff;
    static const bool no_stripe_value = getenv("RLGPU_NO_VALUE_STRIPE") != nullptr;
    if (!no_stripe_value && fused_capable(l) && rows >= 4 * fused::R) {
        // the critic's forward chain per 128-row stripe (ppo_fused.h k_value_stripe): the weights are streamed once per 128 rows instead of once per 32
        if ((rc = refresh_shadows(l))) return rc;
        fused::ValueArgs a{};
        a.obs = obs; a.rows = rows; a.D = l->cfg.obs_size; a.values = values;
        for (int i = 0; i < 4; i++) { a.wf[i] = l->shadows + l->cri.wf16_off[i]; a.bias[i] = l->params + l->cri.b_off[i]; }
        const dim3 grid((rows + fused::R - 1) / fused::R);
        static bool attr[3] = {false, false, false};
        const int which = l->cri.kp[0] == 96 ? 0 : (l->cri.kp[0] == 128 ? 1 : 2);
        const void* fn = which == 0 ? reinterpret_cast<const void*>(&fused::k_value_stripe<96>) : (which == 1 ? reinterpret_cast<const void*>(&fused::k_value_stripe<128>) : reinterpret_cast<const void*>(&fused::k_value_stripe<192>));
        if (!attr[which]) { LCHK(l, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused::SMEM_BYTES)); attr[which] = true; }
        if (which == 0) hipLaunchKernelGGL(fused::k_value_stripe<96>, grid, dim3(512), fused::SMEM_BYTES, l->stream, a);
        else if (which == 1) hipLaunchKernelGGL(fused::k_value_stripe<128>, grid, dim3(512), fused::SMEM_BYTES, l->stream, a);
        else hipLaunchKernelGGL(fused::k_value_stripe<192>, grid, dim3(512), fused::SMEM_BYTES, l->stream, a);
        LCHK(l, hipGetLastError());
        return RLGPU_OK;
    }
    if (rows <= fused_cap && fused_infer_fits(l, l->cri, 1)) return launch_fused_infer(l, l->cri, obs, rows, 1, values, HeadArgs{});
    if (l->cfg.use_bf16) { if ((rc = stage_input16(l, obs, nullptr, rows))) return rc; rc = net_forward16(l, l->cri, l->act16_c, l->act_c.back(), rows); }
    else rc = net_forward(l, l->cri, l->act_c, obs, rows);
    if (rc) return rc;
    LCHK(l, hipMemcpyAsync(values, l->act_c.back(), (size_t)rows * 4, hipMemcpyDeviceToDevice, l->stream));
    return RLGPU_OK;
}

int rlgpu_gae(rlgpu_learner* l, const float* rews, const float* dones, const float* truncs, const float* values, int T, int n,
              float gamma, float lambda, float ret_std, float clip_range, int mode, float* adv, float* targets, float* returns) {
    LCHK(l, hipSetDevice(l->device));
    dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(k_gae, grid, block, 0, l->stream, rews, dones, truncs, values, T, n, gamma, lambda, ret_std, clip_range, mode, adv, targets, returns);
    LCHK(l, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_gae_ragged(rlgpu_learner* l, const float* rews, const float* dones, const float* truncs, const float* values, int n, const int32_t* steps, int players,
                     float gamma, float lambda, float ret_std, float clip_range, int mode, float* adv, float* targets, float* returns) {
    if (!rews || !dones || !values || !steps || !adv || !targets || !returns || n <= 0 || players <= 0 || n % players) { l->err = "rlgpu_gae_ragged: bad argument"; return RLGPU_ERR_ARG; }
    LCHK(l, hipSetDevice(l->device));
    dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(k_gae_ragged, grid, block, 0, l->stream, rews, dones, truncs, values, n, steps, players, gamma, lambda, ret_std, clip_range, mode, adv, targets, returns);
    LCHK(l, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_zero_grads(rlgpu_learner* l) {
    LCHK(l, hipSetDevice(l->device));
    LCHK(l, hipMemsetAsync(l->grads, 0, l->n_total * 4, l->stream));
    return RLGPU_OK;
}

int rlgpu_ppo_minibatch(rlgpu_learner* l, const float* obs, const int32_t* actions, const float* old_logp, const float* adv, const float* targets,
                        const int32_t* idx, int n, float ratio, float* metrics) {
    if (n <= 0 || n > l->cfg.max_rows) { l->err = "minibatch rows out of range (max_rows)"; return RLGPU_ERR_ARG; }
    LCHK(l, hipSetDevice(l->device));
    const int D = l->cfg.obs_size, A = l->cfg.n_actions;
    const bool fast = l->cfg.use_bf16 != 0;
    const float* x = obs;
    int rc;
    const bool use_fused = fast && fused_capable(l);   // (gathers by itself, inside the timed section)
    if (use_fused) {
    } else if (fast) {
        if ((rc = stage_input16(l, obs, idx, n))) return rc;   // gather + bf16 conversion in one pass
    } else if (idx) {
        size_t tot = (size_t)n * D;
        hipLaunchKernelGGL(k_gather_rows, dim3((tot + 255) / 256), dim3(256), 0, l->stream, obs, idx, n, D, l->gathered);
        LCHK(l, hipGetLastError());
        x = l->gathered;
    }
    l->last_flops = 0;
    size_t ev_slot = 0;
    if (l->timing_on) {   // rlgpu_learner_enable_timing (bench, profiling tools): the GEMM section of every minibatch between two events
        if (l->ev_used == l->ev_pool.size()) {
            if (l->ev_pool.size() < 1024) {
                hipEvent_t a, b; LCHK(l, hipEventCreate(&a)); LCHK(l, hipEventCreate(&b));
                l->ev_pool.push_back({a, b}); l->ev_flops.push_back(0.0);
            } else {
                int rc2 = rlgpu_learner_timing_total(l, nullptr, nullptr, nullptr, 0);
                if (rc2) return rc2;
            }
        }
        ev_slot = l->ev_used++;
        l->ev0 = l->ev_pool[ev_slot].first; l->ev1 = l->ev_pool[ev_slot].second;
        LCHK(l, hipEventRecord(l->ev0, l->stream));
    }
    const float inv_t = 1.0f / (l->cfg.temperature > 0 ? l->cfg.temperature : 1.f);
    const int loss_blocks = std::max(1, std::min(2048, (n + 3) / 4));
    const int vloss_blocks = std::max(1, std::min(1024, (n + 255) / 256));
    if (use_fused) {
        if ((rc = fused_minibatch(l, obs, actions, old_logp, adv, targets, idx, n, ratio, metrics))) return rc;
    } else if (fast && stripe_capable(l)) {
        // mlp_stripe.h: forward of both networks in ONE launch (activations in LDS from layer to layer), the two loss kernels, the dX chain of
        // both networks in ONE launch, then the eight dW GEMMs -- the policy's and the critic's on a stream each
        if ((rc = refresh_shadows(l))) return rc;
        if ((rc = stripe_launch(l, false, n))) return rc;
        hipLaunchKernelGGL(k_value_loss, dim3(vloss_blocks), dim3(256), 0, l->stream, (const float*)l->act_c.back(), targets, idx, n, ratio / (float)n,
                           (float*)nullptr, l->dy16[1][l->cri.n_layers - 1], l->cri.kp[l->cri.n_layers], metrics);
        hipLaunchKernelGGL(k_ppo_policy_loss, dim3(loss_blocks), dim3(256), 0, l->stream, (const float*)l->act_p.back(), A, n, A, inv_t, actions, old_logp, adv, idx,
                           l->cfg.clip_range, l->cfg.ent_coef, ratio / (float)n, (float*)nullptr, l->dy16[0][l->pol.n_layers - 1], l->pol.kp[l->pol.n_layers], metrics);
        LCHK(l, hipGetLastError());
        if ((rc = stripe_launch(l, true, n))) return rc;
        const bool two = l->dw_stream[0] != nullptr && l->dw_stream[1] != nullptr && !RLGPU_EXPERIMENT_ENV("RLGPU_ONE_STREAM");
        if (two) {
            LCHK(l, hipEventRecord(l->ev_fork, l->stream));
            for (int w = 0; w < 2; w++) {
                LCHK(l, hipStreamWaitEvent(l->dw_stream[w], l->ev_fork, 0));
                if ((rc = net_dw16(l, w == 0 ? l->pol : l->cri, w == 0 ? l->act16_p : l->act16_c, n, w, l->dw_stream[w]))) return rc;
                LCHK(l, hipEventRecord(l->ev_dw_done[w], l->dw_stream[w]));
                LCHK(l, hipStreamWaitEvent(l->stream, l->ev_dw_done[w], 0));
            }
        } else {
            if ((rc = net_dw16(l, l->pol, l->act16_p, n, 0, l->stream))) return rc;
            if ((rc = net_dw16(l, l->cri, l->act16_c, n, 1, l->stream))) return rc;
        }
    } else if (fast) {
        // The two networks are independent until the optimizer step: the critic's chain goes to a side stream, the policy's stays on the
        // learner's, and they meet again before the closing timing event.  A chain is ~12 short dependent launches; every launch boundary
        // waits for the previous layer's output to leave the per-XCD L2s, and the other chain's kernels fill those gaps.
        if ((rc = refresh_shadows(l))) return rc;
        hipStream_t main_stream = l->stream;
        const bool two = l->side != nullptr && !RLGPU_EXPERIMENT_ENV("RLGPU_ONE_STREAM");
        if (two) { LCHK(l, hipEventRecord(l->ev_fork, main_stream)); LCHK(l, hipStreamWaitEvent(l->side, l->ev_fork, 0)); l->stream = l->side; }
        // critic
        rc = net_forward16(l, l->cri, l->act16_c, l->act_c.back(), n);
        if (!rc) {
            hipLaunchKernelGGL(k_value_loss, dim3(vloss_blocks), dim3(256), 0, l->stream, (const float*)l->act_c.back(), targets, idx, n, ratio / (float)n,
                               (float*)nullptr, l->dy16[1][l->cri.n_layers - 1], l->cri.kp[l->cri.n_layers], metrics);
            rc = net_backward16(l, l->cri, l->act16_c, n, l->dy16[1][l->cri.n_layers - 1], two ? 1 : -1);
        }
        if (two) { l->stream = main_stream; if (!rc) { LCHK(l, hipEventRecord(l->ev_join, l->side)); } }
        if (rc) return rc;
        LCHK(l, hipGetLastError());
        // policy
        if ((rc = net_forward16(l, l->pol, l->act16_p, l->act_p.back(), n))) return rc;
        hipLaunchKernelGGL(k_ppo_policy_loss, dim3(loss_blocks), dim3(256), 0, l->stream, (const float*)l->act_p.back(), A, n, A, inv_t, actions, old_logp, adv, idx,
                           l->cfg.clip_range, l->cfg.ent_coef, ratio / (float)n, (float*)nullptr, l->dy16[0][l->pol.n_layers - 1], l->pol.kp[l->pol.n_layers], metrics);
        LCHK(l, hipGetLastError());
        if ((rc = net_backward16(l, l->pol, l->act16_p, n, l->dy16[0][l->pol.n_layers - 1], two ? 0 : -1))) return rc;
        if (two) {
            LCHK(l, hipStreamWaitEvent(main_stream, l->ev_join, 0));
            LCHK(l, hipStreamWaitEvent(main_stream, l->ev_dw_done[0], 0)); LCHK(l, hipStreamWaitEvent(main_stream, l->ev_dw_done[1], 0));
        }
    } else {
        // critic
        if ((rc = net_forward(l, l->cri, l->act_c, x, n))) return rc;
        hipLaunchKernelGGL(k_value_loss, dim3(vloss_blocks), dim3(256), 0, l->stream, (const float*)l->act_c.back(), targets, idx, n, ratio / (float)n, l->dbuf0,
                           (short*)nullptr, 0, metrics);
        LCHK(l, hipGetLastError());
        if ((rc = net_backward(l, l->cri, l->act_c, x, n, l->dbuf0))) return rc;
        // policy
        if ((rc = net_forward(l, l->pol, l->act_p, x, n))) return rc;
        hipLaunchKernelGGL(k_ppo_policy_loss, dim3(loss_blocks), dim3(256), 0, l->stream, (const float*)l->act_p.back(), A, n, A, inv_t, actions, old_logp, adv, idx,
                           l->cfg.clip_range, l->cfg.ent_coef, ratio / (float)n, l->dbuf0, (short*)nullptr, 0, metrics);
        LCHK(l, hipGetLastError());
        if ((rc = net_backward(l, l->pol, l->act_p, x, n, l->dbuf0))) return rc;
    }
    if (l->timing_on) {
        LCHK(l, hipEventRecord(l->ev1, l->stream));
        l->ev_flops[ev_slot] = l->last_flops;
        l->timed = true;
    }
    return RLGPU_OK;
}

int rlgpu_clip_adam_step(rlgpu_learner* l, float max_norm, float grad_scale) {
    LCHK(l, hipSetDevice(l->device));
    struct Seg { int64_t off, n; float lr; int64_t* step; int slot; } segs[2] = {
        {0, l->pol.n_params, l->cfg.policy_lr, &l->step_p, 0}, {l->pol.n_params, l->cri.n_params, l->cfg.critic_lr, &l->step_c, 1}};
    // fp16 mode: the gradients are loss_scale times too large.  GradScaler::unscale_ BEFORE the clip (the reference clips the scaled gradients and
    // unscales inside step(), PPOLearner.cpp:273-297 -- an effective clip norm of 0.5 / scale), a step with a non-finite gradient
    // norm is skipped for that optimizer and the scale backs off, gradscaler.hpp:162,291.
    // All of it on the device (LsDev, k_ls_decide): the call returns without waiting for the minibatch.
    const LsDev* const ls = l->ls_dev;
    for (auto& s : segs) {
        float* const partial = l->norm_buf + 4 + s.slot * SUMSQ_BLOCKS;
        hipLaunchKernelGGL(k_sumsq, dim3(SUMSQ_BLOCKS), dim3(256), 0, l->stream, (const float*)(l->grads + s.off), s.n, grad_scale, partial, ls);
        hipLaunchKernelGGL(k_sumsq_finish, dim3(1), dim3(1), 0, l->stream, (const float*)partial, SUMSQ_BLOCKS, l->norm_buf + s.slot);
        LCHK(l, hipGetLastError());
    }
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    if (ls) { hipLaunchKernelGGL(k_ls_decide, dim3(1), dim3(1), 0, l->stream, (const float*)l->norm_buf, l->ls_dev, b1, b2); LCHK(l, hipGetLastError()); }
    for (auto& s : segs) {
        float bc1 = 1.f, bc2s = 1.f;
        if (!ls) {      // (fp16 mode: the step counts and bias corrections are the device's)
            (*s.step)++;
            double t = (double)*s.step;
            bc1 = (float)(1.0 - std::pow((double)b1, t));
            bc2s = (float)std::sqrt(1.0 - std::pow((double)b2, t));
        }
        hipLaunchKernelGGL(k_clip_adam, dim3((s.n + 255) / 256), dim3(256), 0, l->stream, l->params + s.off, (const float*)(l->grads + s.off),
                           l->adam_m + s.off, l->adam_v + s.off, s.n, (const float*)(l->norm_buf + s.slot), grad_scale, max_norm, s.lr, b1, b2, eps, bc1, bc2s, ls, s.slot);
        LCHK(l, hipGetLastError());
    }
    l->shadows_dirty = true;
    return RLGPU_OK;
}
int rlgpu_learner_loss_scale(rlgpu_learner* l, float* scale, int* growth_steps, int* skipped_steps) {
    LCHK(l, hipSetDevice(l->device));
    { int rc = ls_pull(l); if (rc) return rc; }
    if (scale) *scale = l->shadows_h ? l->loss_scale : 1.f;
    if (growth_steps) *growth_steps = l->ls_growth; if (skipped_steps) *skipped_steps = l->ls_skipped;
    return RLGPU_OK;
}
int rlgpu_learner_set_lr(rlgpu_learner* l, float plr, float clr) { l->cfg.policy_lr = plr; l->cfg.critic_lr = clr; return RLGPU_OK; }
}  // extern "C"
int rlgpu_internal_policy_net(rlgpu_learner* l, rlinfer::InferNet* net, rlinfer::HeadArgs* head, int deterministic, int n_calls, int max_buf_bytes, void* stream) {
    if (l->cfg.n_actions > 128 || l->pol.n_layers > 9) return RLGPU_ERR_STATE;
    int maxkp = 0;
    for (int i = 0; i < l->pol.n_layers; i++) maxkp = std::max(maxkp, l->pol.kp[i]);
    if (!l->cfg.use_bf16) {
        // exact-parity mode: fp32 operands (wave_infer_f32).  max_buf_bytes < 0 carries the bytes one HALF buffer may take
        // (rlgpu_env.hip lends four of them per wavefront); the fp32 master weights are read as they are.
        const Net& n = l->pol;
        int maxk = 0;
        for (int i = 0; i < n.n_layers; i++) maxk = std::max(maxk, (n.dims[i] + 31) / 32 * 32);
        if (max_buf_bytes >= 0) return RLGPU_ERR_STATE;    // (the caller checks the buffer parts against its row count: f32_part_bytes)
        net->n_layers = n.n_layers; net->D = l->cfg.obs_size; net->ld = maxk + 8; net->fp32 = 1;
        for (int i = 0; i < n.n_layers; i++) {
            net->W[i] = nullptr; net->bias[i] = l->params + n.b_off[i];
            net->K[i] = (n.dims[i] + 31) / 32 * 32; net->N[i] = n.dims[i + 1]; net->Npad[i] = (n.dims[i + 1] + 31) / 32 * 32;
            net->Wf[i] = l->params + n.w_off[i]; net->Kf[i] = n.dims[i];
        }
        const float inv_t = 1.0f / (l->cfg.temperature > 0 ? l->cfg.temperature : 1.f);
        *head = HeadArgs{l->cfg.n_actions, inv_t, deterministic, nullptr, l->cfg.seed_lo, (l->cfg.seed_hi ^ 0x5A3C0DEu) + l->sampler_stream * 0x9E3779B9u, l->call_ctr, nullptr, nullptr, nullptr};
        l->call_ctr += (uint32_t)n_calls;
        (void)stream;
        return RLGPU_OK;
    }
    if (max_buf_bytes < 0) max_buf_bytes = 0x7fffffff;   // (the caller sized for the fp32 mode; checked again by it for bf16)
    if (maxkp > 256) return RLGPU_ERR_STATE;   // wave_infer keeps a layer's 16 K steps in registers
    LCHK(l, hipSetDevice(l->device));
    hipStream_t keep = l->stream;
    l->stream = (hipStream_t)stream;            // the weight copies must be current on the stream the caller launches on
    int rc = refresh_shadows(l);
    l->stream = keep;
    if (rc) return rc;
    const Net& n = l->pol;
    net->n_layers = n.n_layers; net->D = l->cfg.obs_size; net->ld = maxkp + 8; net->fp32 = 0;
    for (int i = 0; i < n.n_layers; i++) {
        net->W[i] = l->shadows + n.wf16_off[i]; net->bias[i] = l->params + n.b_off[i];
        net->K[i] = n.kp[i]; net->N[i] = n.dims[i + 1]; net->Npad[i] = n.kp[i + 1];
        net->Wf[i] = nullptr; net->Kf[i] = n.dims[i];
    }
    const float inv_t = 1.0f / (l->cfg.temperature > 0 ? l->cfg.temperature : 1.f);
    *head = HeadArgs{l->cfg.n_actions, inv_t, deterministic, nullptr, l->cfg.seed_lo, (l->cfg.seed_hi ^ 0x5A3C0DEu) + l->sampler_stream * 0x9E3779B9u, l->call_ctr, nullptr, nullptr, nullptr};
    l->call_ctr += (uint32_t)n_calls;
    return RLGPU_OK;
}
extern "C" {
int rlgpu_learner_refresh_shadows(rlgpu_learner* l) { LCHK(l, hipSetDevice(l->device)); return l->cfg.use_bf16 ? refresh_shadows(l) : RLGPU_OK; }
int rlgpu_learner_set_deterministic(rlgpu_learner* l, int on) { l->deterministic = on != 0; return RLGPU_OK; }
int rlgpu_learner_set_sampler(rlgpu_learner* l, uint32_t stream, uint32_t call_ctr) { l->sampler_stream = stream; l->call_ctr = call_ctr; return RLGPU_OK; }
int rlgpu_learner_get_sampler(rlgpu_learner* l, uint32_t* stream, uint32_t* call_ctr) { if (stream) *stream = l->sampler_stream; if (call_ctr) *call_ctr = l->call_ctr; return RLGPU_OK; }
int rlgpu_allreduce_grads(rlgpu_learner* l, rlgpu_comm* c) {
    if (c && rlgpu_comm_device(c) != l->device) { l->err = "rlgpu_allreduce_grads: the learner lives on device " + std::to_string(l->device) + ", the communicator on device " + std::to_string(rlgpu_comm_device(c)); return RLGPU_ERR_ARG; }
    LCHK(l, hipSetDevice(l->device));
    int rc = rlgpu_comm_allreduce_f32(c, l->grads, l->n_total, (void*)l->stream);
    if (rc) l->err = rlgpu_comm_last_error(c);
    return rc;
}
int rlgpu_learner_sync_from_rank0(rlgpu_learner* l, rlgpu_comm* c) {
    if (!c) return RLGPU_OK;
    LCHK(l, hipSetDevice(l->device));
    for (float* p : {l->params, l->adam_m, l->adam_v}) { int rc = rlgpu_comm_broadcast(c, p, l->n_total * 4, 0, (void*)l->stream); if (rc) { l->err = rlgpu_comm_last_error(c); return rc; } }
    { int rc0 = ls_pull(l); if (rc0) return rc0; }
    int64_t* steps = nullptr;
    LCHK(l, hipMalloc(&steps, 16));
    struct Free { int64_t* p; ~Free() { (void)hipFree(p); } } free_steps{steps};   // (released on every way out of the function, the LCHK returns included)
    const int64_t h[2] = {l->step_p, l->step_c};
    LCHK(l, hipMemcpyAsync(steps, h, 16, hipMemcpyHostToDevice, l->stream));
    int rc = rlgpu_comm_broadcast(c, steps, 16, 0, (void*)l->stream);
    int64_t g[2] = {0, 0};
    if (!rc) { LCHK(l, hipMemcpyAsync(g, steps, 16, hipMemcpyDeviceToHost, l->stream)); LCHK(l, hipStreamSynchronize(l->stream)); l->step_p = g[0]; l->step_c = g[1]; rc = ls_push(l); }
    if (rc) { l->err = rlgpu_comm_last_error(c); return rc; }
    l->shadows_dirty = true;
    return RLGPU_OK;
}
int rlgpu_learner_param_checksum(rlgpu_learner* l, uint64_t* out) {
    if (!out) return RLGPU_ERR_ARG;
    LCHK(l, hipSetDevice(l->device));
    std::vector<uint32_t> h((size_t)l->n_total);
    LCHK(l, hipMemcpyAsync(h.data(), l->params, (size_t)l->n_total * 4, hipMemcpyDeviceToHost, l->stream));
    LCHK(l, hipStreamSynchronize(l->stream));
    uint64_t a = 0xcbf29ce484222325ull;   // FNV-1a over the words
    for (uint32_t w : h) { a ^= w; a *= 0x100000001b3ull; }
    *out = a;
    return RLGPU_OK;
}
int rlgpu_learner_replicas_equal(rlgpu_learner* l, rlgpu_comm* c, int* equal_out) {
    if (!equal_out) return RLGPU_ERR_ARG;
    *equal_out = 1;
    if (!c) return RLGPU_OK;
    uint64_t mine = 0;
    int rc = rlgpu_learner_param_checksum(l, &mine);
    if (rc) return rc;
    uint64_t* d = nullptr;
    LCHK(l, hipMalloc(&d, 8));
    LCHK(l, hipMemcpyAsync(d, &mine, 8, hipMemcpyHostToDevice, l->stream));
    rc = rlgpu_comm_broadcast(c, d, 8, 0, (void*)l->stream);
    uint64_t root = 0;
    if (!rc) { LCHK(l, hipMemcpyAsync(&root, d, 8, hipMemcpyDeviceToHost, l->stream)); LCHK(l, hipStreamSynchronize(l->stream)); }
    (void)hipFree(d);
    if (rc) { l->err = rlgpu_comm_last_error(c); return rc; }
    *equal_out = root == mine ? 1 : 0;
    return RLGPU_OK;
}
int rlgpu_learner_set_temperature(rlgpu_learner* l, float t) { if (!(t > 0)) return RLGPU_ERR_ARG; l->cfg.temperature = t; return RLGPU_OK; }
int rlgpu_learner_sync(rlgpu_learner* l) { LCHK(l, hipSetDevice(l->device)); LCHK(l, hipStreamSynchronize(l->stream)); return RLGPU_OK; }
int rlgpu_learner_last_gemm(rlgpu_learner* l, float* ms, double* flops) {
    if (!l->timed) return RLGPU_ERR_STATE;
    LCHK(l, hipEventSynchronize(l->ev1));
    LCHK(l, hipEventElapsedTime(ms, l->ev0, l->ev1));
    *flops = l->last_flops;
    return RLGPU_OK;
}

int rlgpu_learner_enable_timing(rlgpu_learner* l, int on) { l->timing_on = on != 0; if (!on) l->timed = false; return RLGPU_OK; }
int rlgpu_learner_timing_total(rlgpu_learner* l, float* total_ms, double* total_flops, int* calls, int reset) {
    LCHK(l, hipSetDevice(l->device));
    if (l->ev_used) LCHK(l, hipEventSynchronize(l->ev_pool[l->ev_used - 1].second));   // the newest pair: whatever stream it was recorded on
    for (size_t i = 0; i < l->ev_used; i++) {
        float ms = 0.f; LCHK(l, hipEventElapsedTime(&ms, l->ev_pool[i].first, l->ev_pool[i].second));
        l->acc_ms += ms; l->acc_flops += l->ev_flops[i]; l->acc_calls++;
    }
    l->ev_used = 0;
    if (total_ms) *total_ms = (float)l->acc_ms;
    if (total_flops) *total_flops = l->acc_flops;
    if (calls) *calls = l->acc_calls;
    if (reset) { l->acc_ms = 0; l->acc_flops = 0; l->acc_calls = 0; }
    return RLGPU_OK;
}

int rlgpu_shuffler_create(rlgpu_shuffler** out, uint32_t seed) {
    if (!out) return RLGPU_ERR_ARG;
    *out = new rlgpu_shuffler{std::default_random_engine(seed)};
    return RLGPU_OK;
}
void rlgpu_shuffler_destroy(rlgpu_shuffler* s) { delete s; }
int rlgpu_shuffler_next(rlgpu_shuffler* s, int64_t n, int64_t* perm) {
    if (!s || n < 0 || !perm) return RLGPU_ERR_ARG;
    std::iota(perm, perm + n, (int64_t)0);
    std::shuffle(perm, perm + n, s->rng);
    return RLGPU_OK;
}
int rlgpu_shuffler_next_i32(rlgpu_shuffler* s, int64_t n, int32_t* perm) {
    if (!s || n < 0 || n > 0x7fffffffLL || !perm) return RLGPU_ERR_ARG;
    s->scratch.resize((size_t)n);
    int rc = rlgpu_shuffler_next(s, n, s->scratch.data());   // the 64-bit draw, narrowed: the engine is consumed exactly as by rlgpu_shuffler_next
    if (rc) return rc;
    for (int64_t i = 0; i < n; i++) perm[i] = (int32_t)s->scratch[(size_t)i];
    return RLGPU_OK;
}
int rlgpu_shuffler_get_state(const rlgpu_shuffler* s, char* buf, int cap) {
    if (!s || !buf) return RLGPU_ERR_ARG;
    std::ostringstream o; o << s->rng;
    const std::string t = o.str();
    if ((int)t.size() + 1 > cap) return RLGPU_ERR_ARG;
    memcpy(buf, t.c_str(), t.size() + 1);
    return RLGPU_OK;
}
int rlgpu_shuffler_set_state(rlgpu_shuffler* s, const char* buf) {
    if (!s || !buf) return RLGPU_ERR_ARG;
    std::istringstream i{std::string(buf)}; i >> s->rng;
    return i.fail() ? RLGPU_ERR_ARG : RLGPU_OK;
}
int rlgpu_shuffler_next_rows(rlgpu_shuffler* s, int T, int n_agents, int32_t* rows) {
    if (!s || T <= 0 || n_agents <= 0 || !rows) return RLGPU_ERR_ARG;
    const int64_t B = (int64_t)T * n_agents;
    s->scratch.resize((size_t)B);
    int rc = rlgpu_shuffler_next(s, B, s->scratch.data());
    if (rc) return rc;
    for (int64_t i = 0; i < B; i++) { const int64_t p = s->scratch[(size_t)i]; rows[i] = (int32_t)((p % T) * n_agents + p / T); }
    return RLGPU_OK;
}

// ---- experience FIFO (host bookkeeping only; the rows themselves stay in the caller's device slots) ----
int rlgpu_expbuf_create(rlgpu_expbuf** out, int64_t max_rows, int T, int n_agents) {
    if (!out || max_rows <= 0 || T <= 0 || n_agents <= 0) return RLGPU_ERR_ARG;
    rlgpu_expbuf* b = new rlgpu_expbuf();
    b->max_rows = max_rows; b->T = T; b->n = n_agents; b->B = (int64_t)T * n_agents;
    b->n_slots = (int)((max_rows + b->B - 1) / b->B) + 1;
    *out = b;
    return RLGPU_OK;
}
int rlgpu_expbuf_create_ragged(rlgpu_expbuf** out, int64_t max_rows, int T_cap, int n_agents, int64_t min_rows_per_iteration) {
    if (!out || max_rows <= 0 || T_cap <= 0 || n_agents <= 0 || min_rows_per_iteration <= 0) return RLGPU_ERR_ARG;
    rlgpu_expbuf* b = new rlgpu_expbuf();
    b->max_rows = max_rows; b->T = T_cap; b->n = n_agents; b->B = (int64_t)T_cap * n_agents;
    b->n_slots = (int)((max_rows + min_rows_per_iteration - 1) / min_rows_per_iteration) + 1;
    if (b->n_slots > 16) { delete b; return RLGPU_ERR_ARG; }   // (k_map_rows takes the chunk table by value)
    *out = b;
    return RLGPU_OK;
}
void rlgpu_expbuf_destroy(rlgpu_expbuf* b) { delete b; }
int rlgpu_expbuf_num_slots(const rlgpu_expbuf* b) { return b ? b->n_slots : 0; }
int64_t rlgpu_expbuf_size(const rlgpu_expbuf* b) {
    int64_t n = 0;
    if (b) for (auto& c : b->chunks) n += c.rows - c.skip;
    return n;
}
static int expbuf_submit(rlgpu_expbuf* b, int64_t rows, int64_t keep_last, std::vector<int64_t>&& off, int* slot_out) {
    // an addition larger than the buffer keeps its LAST max_rows rows (ExperienceBuffer.cpp:32-35)
    int64_t add = rows, new_skip = 0;
    if (keep_last > 0 && add > keep_last) add = keep_last;
    if (add > b->max_rows) add = b->max_rows;
    new_skip = rows - add;
    // shift left by the overflow (ExperienceBuffer.cpp:37-58): drop the oldest rows
    int64_t overflow = std::max<int64_t>(rlgpu_expbuf_size(b) + add - b->max_rows, 0);
    while (overflow > 0 && !b->chunks.empty()) {
        rlgpu_expbuf::Chunk& c = b->chunks.front();
        int64_t have = c.rows - c.skip;
        if (have <= overflow) { overflow -= have; b->chunks.erase(b->chunks.begin()); }
        else { c.skip += overflow; overflow = 0; }
    }
    while (!b->chunks.empty() && b->chunks.front().rows == b->chunks.front().skip) b->chunks.erase(b->chunks.begin());   // (an iteration of no rows)
    int slot = -1;
    for (int s = 0; s < b->n_slots && slot < 0; s++) {
        bool used = false;
        for (auto& c : b->chunks) used = used || c.slot == s;
        if (!used) slot = s;
    }
    if (slot < 0) return RLGPU_ERR_STATE;
    b->chunks.push_back({slot, new_skip, rows, std::move(off)});
    *slot_out = slot;
    return RLGPU_OK;
}
int rlgpu_expbuf_submit(rlgpu_expbuf* b, int* slot_out) {
    if (!b || !slot_out) return RLGPU_ERR_ARG;
    return expbuf_submit(b, b->B, 0, {}, slot_out);
}
int rlgpu_expbuf_submit_ragged(rlgpu_expbuf* b, const int32_t* agent_steps, int64_t keep_last, int* slot_out) {
    if (!b || !slot_out || !agent_steps || keep_last < 0) return RLGPU_ERR_ARG;
    std::vector<int64_t> off((size_t)b->n + 1, 0);
    for (int a = 0; a < b->n; a++) {
        if (agent_steps[a] < 0 || agent_steps[a] > b->T) return RLGPU_ERR_ARG;
        off[(size_t)a + 1] = off[(size_t)a] + agent_steps[a];
    }
    const int64_t rows = off[(size_t)b->n];
    return expbuf_submit(b, rows, keep_last, std::move(off), slot_out);
}
// logical FIFO row p -> device row
static inline int32_t expbuf_row(const rlgpu_expbuf* b, const std::vector<int64_t>& start, int64_t p) {
    size_t c = (size_t)(std::upper_bound(start.begin(), start.end(), p) - start.begin()) - 1;
    const rlgpu_expbuf::Chunk& ch = b->chunks[c];
    const int64_t a = p - start[c] + ch.skip;
    if (ch.off.empty()) return (int32_t)((int64_t)ch.slot * b->B + (a % b->T) * b->n + a / b->T);
    const size_t ag = (size_t)(std::upper_bound(ch.off.begin(), ch.off.end(), a) - ch.off.begin()) - 1;   // the trajectory that owns row a (empty ones own none)
    return (int32_t)((int64_t)ch.slot * b->B + (a - ch.off[ag]) * b->n + (int64_t)ag);
}
int rlgpu_expbuf_shuffled_rows(rlgpu_expbuf* b, rlgpu_shuffler* s, int32_t* rows_out) {
    if (!b || !s || !rows_out) return RLGPU_ERR_ARG;
    const int64_t cur = rlgpu_expbuf_size(b);
    if ((int64_t)b->n_slots * b->B > 0x7fffffffLL) return RLGPU_ERR_ARG;
    s->scratch.resize((size_t)cur);
    int rc = rlgpu_shuffler_next(s, cur, s->scratch.data());
    if (rc) return rc;
    // logical row i of the FIFO -> (chunk, agent-major index a) -> device row slot * B + t * n + agent
    std::vector<int64_t> start(b->chunks.size() + 1, 0);
    for (size_t c = 0; c < b->chunks.size(); c++) start[c + 1] = start[c] + (b->chunks[c].rows - b->chunks[c].skip);
    for (int64_t i = 0; i < cur; i++) rows_out[i] = expbuf_row(b, start, s->scratch[(size_t)i]);
    return RLGPU_OK;
}
int rlgpu_expbuf_map_rows(rlgpu_expbuf* b, const int32_t* perm, int64_t n, int32_t* rows_out) {
    if (!b || !perm || !rows_out || n != rlgpu_expbuf_size(b) || (int64_t)b->n_slots * b->B > 0x7fffffffLL) return RLGPU_ERR_ARG;
    std::vector<int64_t> start(b->chunks.size() + 1, 0);
    for (size_t c = 0; c < b->chunks.size(); c++) start[c + 1] = start[c] + (b->chunks[c].rows - b->chunks[c].skip);
    for (int64_t i = 0; i < n; i++) { if (perm[i] < 0 || perm[i] >= n) return RLGPU_ERR_ARG; rows_out[i] = expbuf_row(b, start, perm[i]); }
    return RLGPU_OK;
}
int rlgpu_expbuf_map_rows_dev(rlgpu_expbuf* b, const int32_t* perm_dev, int64_t n, const int32_t* traj_off_dev, int32_t* rows_dev, void* stream) {
    if (!b || !perm_dev || !rows_dev || !traj_off_dev || n != rlgpu_expbuf_size(b) || b->chunks.size() > 16 || (int64_t)b->n_slots * b->B > 0x7fffffffLL) return RLGPU_ERR_ARG;
    ExpChunks ch{}; ch.count = (int)b->chunks.size(); ch.start[0] = 0;
    for (int c = 0; c < ch.count; c++) {
        if (b->chunks[(size_t)c].off.empty()) return RLGPU_ERR_STATE;   // (a lockstep iteration in the FIFO: its offsets are not on the device; map on the host)
        ch.slot[c] = b->chunks[(size_t)c].slot; ch.skip[c] = b->chunks[(size_t)c].skip; ch.start[c + 1] = ch.start[c] + (b->chunks[(size_t)c].rows - b->chunks[(size_t)c].skip);
    }
    if (n == 0) return RLGPU_OK;
    hipLaunchKernelGGL(k_map_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, perm_dev, (long long)n, ch, traj_off_dev, b->n, (long long)b->B, rows_dev);
    return hipGetLastError() == hipSuccess ? RLGPU_OK : RLGPU_ERR_HIP;
}
int rlgpu_traj_offsets(const int32_t* steps_dev, int n_agents, int players, int32_t* off_dev, void* stream) {
    if (!steps_dev || !off_dev || n_agents <= 0 || players <= 0) return RLGPU_ERR_ARG;
    hipLaunchKernelGGL(k_traj_offsets, dim3(1), dim3(1024), 0, (hipStream_t)stream, steps_dev, n_agents, players, off_dev);
    return hipGetLastError() == hipSuccess ? RLGPU_OK : RLGPU_ERR_HIP;
}

}  // extern "C"
