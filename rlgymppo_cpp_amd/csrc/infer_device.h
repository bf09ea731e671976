// infer_device.h -- device code shared by the learner's inference kernels (rlgpu_learn.hip) and the fused collection kernel of the
// env (rlgpu_env.hip): the policy head (softmax / clamp / sampler / log-prob) and the description of an MLP as the MFMA kernels read
// it.  Device-only (hipcc); reference semantics: DiscretePolicy::GetActionProbs / GetAction (PRIV/PPO/DiscretePolicy.cpp:44-62).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include "rl_math.h"

namespace rlinfer {
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
__device__ __forceinline__ short to_bf16(float f) {   // the hardware conversion, exactly rlgpu_learn.hip's f2bf
    __hip_bfloat16 h = __float2bfloat16(f);
    return *reinterpret_cast<short*>(&h);
}
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

}  // namespace rlinfer

namespace rlinfer {
// one wave per row: probs = clamp(softmax(logits / T), 1e-11, 1); action = argmax(p / q) or argmax(p); logp = log p[a]
struct HeadArgs {
    int A; float inv_temp; int deterministic; const float* noise;
    uint32_t seed_lo, seed_hi, call_ctr;
    int32_t* actions; float* logp; float* probs_out;
};

// NR rows at once: the head is a chain of cross-lane steps (two reductions, an arg-max butterfly), each a ~100-cycle round trip;
// rows are independent, so a wavefront that owns several interleaves them and the round trips overlap.  Per row the arithmetic
// is the same for every NR.
template <int NR>
__device__ __forceinline__ void policy_head_rows(const float* const (&z)[NR], const int (&row)[NR], int lane, const HeadArgs& h, int* picked = nullptr) {
    // this code is compiled into two translation units with different -ffp-contract settings and must round identically in both (the
    // fused collection kernel promises the batched kernels' bits): pin the setting here
#pragma clang fp contract(fast)
    const int A = h.A; const float inv_temp = h.inv_temp; const int deterministic = h.deterministic; const float* noise = h.noise;
    int32_t* actions = h.actions; float* logp = h.logp; float* probs_out = h.probs_out;
    const bool in0 = lane < A, in1 = (lane + 64) < A;
    float v0[NR], v1[NR], mx[NR], e0[NR], e1[NR], sum[NR], p0[NR], p1[NR];
#pragma unroll
    for (int n = 0; n < NR; n++) {
        v0[n] = in0 ? z[n][lane] * inv_temp : -INFINITY;
        v1[n] = in1 ? z[n][lane + 64] * inv_temp : -INFINITY;
        mx[n] = fmaxf(v0[n], v1[n]);
    }
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int n = 0; n < NR; n++) mx[n] = fmaxf(mx[n], __shfl_xor(mx[n], o, 64));
#pragma unroll
    for (int n = 0; n < NR; n++) {
        e0[n] = in0 ? expf(v0[n] - mx[n]) : 0.f; e1[n] = in1 ? expf(v1[n] - mx[n]) : 0.f;
        sum[n] = e0[n] + e1[n];
    }
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int n = 0; n < NR; n++) sum[n] += __shfl_xor(sum[n], o, 64);
#pragma unroll
    for (int n = 0; n < NR; n++) {
        p0[n] = fminf(fmaxf(e0[n] / sum[n], 1e-11f), 1.f); p1[n] = fminf(fmaxf(e1[n] / sum[n], 1e-11f), 1.f);
        if (probs_out) {
            if (in0) probs_out[(size_t)row[n] * A + lane] = p0[n];
            if (in1) probs_out[(size_t)row[n] * A + lane + 64] = p1[n];
        }
    }
    if (!actions) return;
    float best[NR]; int bi[NR];
#pragma unroll
    for (int n = 0; n < NR; n++) {
        float s0, s1;
        if (deterministic) { s0 = p0[n]; s1 = p1[n]; }
        else {
            float q0, q1;
            if (noise) { q0 = in0 ? noise[(size_t)row[n] * A + lane] : 1.f; q1 = in1 ? noise[(size_t)row[n] * A + lane + 64] : 1.f; }
            else {
                uint32_t r[4];
                rlg::philox4(h.seed_lo, h.seed_hi, (uint32_t)row[n], h.call_ctr, (uint32_t)lane, r);
                // q ~ Exp(1): -log(1 - u), u in [0,1)
                q0 = -logf(1.f - rlg::u32_to_unit(r[0])); q1 = -logf(1.f - rlg::u32_to_unit(r[1]));
                q0 = fmaxf(q0, 1e-30f); q1 = fmaxf(q1, 1e-30f);
            }
            s0 = p0[n] / q0; s1 = p1[n] / q1;
        }
        if (!in0) s0 = -INFINITY;
        if (!in1) s1 = -INFINITY;
        best[n] = s0; bi[n] = lane;
        if (s1 > best[n]) { best[n] = s1; bi[n] = lane + 64; }
    }
    // argmax with lowest-index tie break (torch.argmax / max semantics)
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int n = 0; n < NR; n++) {
            float ob = __shfl_xor(best[n], o, 64); int oi = __shfl_xor(bi[n], o, 64);
            if (ob > best[n] || (ob == best[n] && oi < bi[n])) { best[n] = ob; bi[n] = oi; }
        }
#pragma unroll
    for (int n = 0; n < NR; n++) {
        float pa = (bi[n] < 64) ? __shfl(p0[n], bi[n], 64) : __shfl(p1[n], bi[n] - 64, 64);
        if (lane == 0) { actions[row[n]] = bi[n]; logp[row[n]] = deterministic ? 0.f : logf(pa); }
        if (picked) picked[n] = bi[n];   // uniform over the wavefront after the butterfly
    }
}
}  // namespace rlinfer

// ---- an MLP as the MFMA kernels read it, and the forward pass of a few rows on ONE wavefront -------------------------------
namespace rlinfer {
struct InferNet {
    int n_layers, D, ld;             // ld = LDS activation row in elements: max K + 8
    const short* W[9];               // fragment-ordered bf16 weights (k_weight_frags): 1 KB per (32-column block, 16-deep K step)
    const float* bias[9];
    int K[9], N[9], Npad[9];         // layer i: K = padded inputs, N = outputs, Npad = outputs padded to 32
    // exact-parity mode (use_bf16 = 0): the fp32 master weights as nn.Linear keeps them, W[out][in] row-major, true widths
    int fp32;                        // 1: wave_infer_f32 (v_mfma_f32_32x32x2_f32 on fp32 operands), 0: wave_infer (bf16 operands)
    const float* Wf[9]; int Kf[9];   // layer i: weights, true number of inputs
};
constexpr int LOGIT_LD = 132;        // fp32 logits row in LDS (n_actions <= 128)
constexpr int WAVE_ROWS = 16;        // rows one wavefront infers at most: they sit in the first 8 or 16 rows of a 32-row MFMA tile
__host__ __device__ constexpr int wave_buf_bytes(int rows, int ld) { return (rows * ld * 2 > rows * LOGIT_LD * 4) ? rows * ld * 2 : rows * LOGIT_LD * 4; }
// the head of R rows in batches of at most 8 (policy_head_rows keeps ~10 registers per row)
template <int R>
__device__ __forceinline__ void policy_head_batched(const float* const (&z)[R], const int (&row)[R], int lane, const HeadArgs& h, int (&picked)[R]) {
    if constexpr (R <= 8) policy_head_rows<R>(z, row, lane, h, picked);
    else {
        constexpr int R1 = R - 8;
        const float* za[8]; int ra[8], pa[8]; const float* zb[R1]; int rb[R1], pb[R1];
#pragma unroll
        for (int r = 0; r < 8; r++) { za[r] = z[r]; ra[r] = row[r]; }
#pragma unroll
        for (int r = 0; r < R1; r++) { zb[r] = z[8 + r]; rb[r] = row[8 + r]; }
        policy_head_rows<8>(reinterpret_cast<const float* const (&)[8]>(za), ra, lane, h, pa);
        policy_head_rows<R1>(reinterpret_cast<const float* const (&)[R1]>(zb), rb, lane, h, pb);
#pragma unroll
        for (int r = 0; r < 8; r++) picked[r] = pa[r];
#pragma unroll
        for (int r = 0; r < R1; r++) picked[8 + r] = pb[r];
    }
}

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One 32-column block of a layer with NK K steps, guard-free (see wave_infer): weight fragments, and the MFMA chain with the A operand
// read four steps ahead into four named registers.
template <int NK>
__device__ __forceinline__ void fetch_block(bf16x8 (&b)[16], const short* w) {
#pragma unroll
    for (int j = 0; j < NK; j++) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)j * 512);
}
template <int NK>
__device__ __forceinline__ void mma_block(const short* arow, const bf16x8 (&b)[16], f32x16& acc) {
#define RLINFER_A(J) (*reinterpret_cast<const bf16x8*>(arow + ((J) < NK ? (J) : 0) * 16))
#define RLINFER_STEP(J, AREG)                                                                    \
    if constexpr ((J) < NK) {                                                                    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AREG, b[J], acc, 0, 0, 0);                 \
        if constexpr ((J) + 4 < NK) AREG = RLINFER_A((J) + 4);                                   \
        __builtin_amdgcn_sched_barrier(0);   /* keep the read four steps ahead: the scheduler sinks it to its use otherwise */ \
    }
    bf16x8 a0 = RLINFER_A(0), a1 = RLINFER_A(1), a2 = RLINFER_A(2), a3 = RLINFER_A(3);
    __builtin_amdgcn_sched_barrier(0);
    RLINFER_STEP(0, a0) RLINFER_STEP(1, a1) RLINFER_STEP(2, a2) RLINFER_STEP(3, a3)
    RLINFER_STEP(4, a0) RLINFER_STEP(5, a1) RLINFER_STEP(6, a2) RLINFER_STEP(7, a3)
    RLINFER_STEP(8, a0) RLINFER_STEP(9, a1) RLINFER_STEP(10, a2) RLINFER_STEP(11, a3)
    RLINFER_STEP(12, a0) RLINFER_STEP(13, a1) RLINFER_STEP(14, a2) RLINFER_STEP(15, a3)
#undef RLINFER_STEP
#undef RLINFER_A
}
// any other depth (<= 16 steps): guarded, slower
__device__ __forceinline__ void fetch_block_any(bf16x8 (&b)[16], const short* w, int nk) {
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < nk) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)j * 512);
}
__device__ __forceinline__ void mma_block_any(const short* arow, const bf16x8 (&b)[16], f32x16& acc, int nk) {
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < nk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + j * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc, 0, 0, 0);
        }
}
// the K depths that occur: 16 (256-wide hidden layers), and the padded observation widths 96 / 128 / 192 / 224 (89, 127, 165, 203 floats)
#define RLINFER_DISPATCH_NK(nk, CALL, FALLBACK) \
    switch (nk) { case 16: CALL(16); break; case 6: CALL(6); break; case 8: CALL(8); break; case 12: CALL(12); break; case 14: CALL(14); break; default: FALLBACK; }

// Policy forward + head for rows row0 .. row0 + R - 1 (R <= 16) whose fp32 observations are obs[r * D + c], by one wavefront.
// Same operand values, accumulation order, bias / ReLU / bf16 rounding and head code as k_mlp_infer (rlgpu_learn.hip): the logits and
// the sampled actions are those of a batched call.  Only the first n_rows (>= 1) of the R rows exist: the others redo the
// last real one (same values, same stores).  buf0 / buf1: LDS, wave_buf_bytes(R, net.ld) each.  picked[r] = the action.
template <int R>
__device__ __forceinline__ void wave_infer(const InferNet& net, const HeadArgs& head, const float* obs, int row0, int n_rows, short* buf0, short* buf1, int lane, int (&picked)[R],
                                           unsigned long long* prof_split = nullptr) {
    static_assert(R <= WAVE_ROWS, "a wavefront infers at most 16 rows");
    constexpr int CHUNK = 16;
    short* in = buf0; short* out = buf1;
    const int ld = net.ld;
    for (int r = 0; r < R; r++) {
        const int rr = r < n_rows ? r : n_rows - 1;
        for (int c = lane; c < net.K[0]; c += 64) in[r * ld + c] = to_bf16(c < net.D ? obs[(size_t)rr * net.D + c] : 0.f);
    }
    wave_fence();
    auto n_blocks = [&](int i) { return (i == net.n_layers - 1) ? (net.N[i] + 31) / 32 : net.Npad[i] / 32; };
    // Weight fragments of one 32-column block (K <= 256: at most CHUNK steps).  The compiler tracks outstanding loads through straight-line
    // code only -- behind per-step `if (step < nk)` guards it waited for EVERY load (s_waitcnt vmcnt(0)) in front of every MFMA, which
    // turned the prefetch of the next block into a stall of the current one.  So: one guard-free path for full-depth blocks (K = 256,
    // the hidden layers and the head, 19 of the 27 blocks of the 256x3 policy), a guarded one for the rest.
    auto fetch = [&](bf16x8 (&b)[CHUNK], int i, int cb) {
        const int nk = net.K[i] / 16;
        const short* w = net.W[i] + ((size_t)cb * nk * 64 + lane) * 8;
#define RLINFER_FETCH(NK) fetch_block<NK>(b, w)
        RLINFER_DISPATCH_NK(nk, RLINFER_FETCH, fetch_block_any(b, w, nk))
#undef RLINFER_FETCH
    };
    // the bias of a block is asked for together with its weights
    auto fetch_bias = [&](int i, int cb) { const int col = cb * 32 + (lane & 31); return (col < net.N[i]) ? net.bias[i][col] : 0.f; };
    // (Tried and dropped: two blocks of look-ahead in three fixed-role buffers -- no copies, no spills -- 79 K -> 97 K cycles per step:
    // the stream is not waiting on look-ahead depth.)
    bf16x8 bnext[CHUNK];
    fetch(bnext, 0, 0);
    float bias_next = fetch_bias(0, 0);
    for (int i = 0; i < net.n_layers; i++) {
        const bool last = (i == net.n_layers - 1);
        const int N = net.N[i], nk = net.K[i] / 16, nblk = n_blocks(i);
        float* const logits = reinterpret_cast<float*>(out);
        // tile rows >= R do not exist: their lanes re-read row 0 (rows of an MFMA are independent and only rows < R are stored), which
        // keeps the operand loads free of exec masking
        const short* arow = in + ((lane & 31) < R ? (lane & 31) : 0) * ld + 8 * (lane >> 5);
        for (int cb = 0; cb < nblk; cb++) {
            bf16x8 b[CHUNK];
#pragma unroll
            for (int j = 0; j < CHUNK; j++) b[j] = bnext[j];
            const float bias = bias_next;
            // the next block's (or the next layer's first block's) weights do not depend on the activations: ask for them now; after
            // the last block of the last layer the current block is simply asked for again
            int ni = i, ncb = cb + 1;
            if (ncb >= nblk) { ni = i + 1; ncb = 0; }
            if (ni >= net.n_layers) { ni = i; ncb = cb; }
            fetch(bnext, ni, ncb);
            bias_next = fetch_bias(ni, ncb);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0.f;
#define RLINFER_MMA(NK) mma_block<NK>(arow, b, acc)
            RLINFER_DISPATCH_NK(nk, RLINFER_MMA, mma_block_any(arow, b, acc, nk))
#undef RLINFER_MMA
            // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): rows 0..7 are registers 0..3, rows 8..15 registers 4..7
            const int col = cb * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < (R <= 8 ? 4 : 8); r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= R) continue;
                const float v = acc[r] + bias;
                if (last) logits[row * LOGIT_LD + col] = v;
                else out[row * ld + col] = (col < N) ? to_bf16(fmaxf(v, 0.f)) : (short)0;
            }
        }
        wave_fence();
        short* t = in; in = out; out = t;
    }
#ifdef RLG_TICK_PROFILE
    if (prof_split) *prof_split = __builtin_amdgcn_s_memtime();   // profiler build: the MLP ends here, the head begins
#endif
    const float* logits = reinterpret_cast<const float*>(in);
    const float* zs[R]; int rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int rr = r < n_rows ? r : n_rows - 1; zs[r] = logits + rr * LOGIT_LD; rows[r] = row0 + rr; }
    const float* const (&zc)[R] = reinterpret_cast<const float* const (&)[R]>(zs);
    policy_head_batched<R>(zc, rows, lane, head, picked);
}
// ---- the same forward pass in fp32: the exact-parity mode of the fused collection ------------------------------------------------------
// rlgpu_policy_act with use_bf16 = 0 runs every layer through k_gemm<false> (rlgpu_learn.hip): v_mfma_f32_32x32x2_f32 over k = 0, 2, 4, ...
// (lane l holds A[row l & 31][k + (l >> 5)] and B[k + (l >> 5)][col l & 31]), operands zero-padded to a multiple of 32 inputs, bias added
// and ReLU applied to the fp32 accumulator.  A row's value depends on nothing but its own operands and that order, so the same instruction
// sequence on one wavefront gives the batched call's bits.  Activations stay fp32 in LDS; a buffer is two or three PARTS (the step kernels lend
// the TickWork areas of different envs) of ceil(R / NP) rows each.
struct F32Buf { float* part[3]; };   // NP = 2 or 3 parts of ceil(R / NP) rows each
template <int R, int NP>
__device__ __forceinline__ float* f32_row(const F32Buf& b, int r, int ld) {   // (selects, not an indexed load: the three pointers stay in registers)
    constexpr int RP = (R + NP - 1) / NP;
    if (NP == 2) return r < RP ? b.part[0] + r * ld : b.part[1] + (r - RP) * ld;
    return r < RP ? b.part[0] + r * ld : (r < 2 * RP ? b.part[1] + (r - RP) * ld : b.part[2] + (r - 2 * RP) * ld);
}
__host__ __device__ constexpr int f32_part_bytes(int rows, int np, int ld) { return ((rows + np - 1) / np) * (ld > LOGIT_LD ? ld : LOGIT_LD) * 4; }

template <int R, int NP>
__device__ __forceinline__ void wave_infer_f32(const InferNet& net, const HeadArgs& head, const float* obs, int row0, int n_rows, F32Buf in, F32Buf out, int lane, int (&picked)[R]) {
    static_assert(R <= WAVE_ROWS, "a wavefront infers at most 16 rows");
    constexpr int CH = 16;                               // MFMA steps (2 inputs each) whose B operands are in flight together
    const int ld = net.ld;
    for (int r = 0; r < R; r++) {
        const int rr = r < n_rows ? r : n_rows - 1;
        float* dst = f32_row<R, NP>(in, r, ld);
        for (int c = lane; c < net.K[0]; c += 64) dst[c] = c < net.D ? obs[(size_t)rr * net.D + c] : 0.f;
    }
    wave_fence();
    const int kh = lane >> 5, cl = lane & 31;
    for (int i = 0; i < net.n_layers; i++) {
        const bool last = (i == net.n_layers - 1);
        const int N = net.N[i], Kt = net.Kf[i];
        const int steps = ((Kt + 31) / 32) * 16;         // k_gemm pads the reduction to whole 32-wide tiles (zeros: they leave the accumulator as it is)
        const int nblk = last ? (N + 31) / 32 : net.Npad[i] / 32;
        const float* arow = f32_row<R, NP>(in, cl < R ? cl : 0, ld);    // tile rows >= R re-read row 0 (independent, never stored)
        for (int cb = 0; cb < nblk; cb++) {
            const int col = cb * 32 + cl;
            const float* wrow = net.Wf[i] + (size_t)(col < N ? col : 0) * Kt;
            const bool col_ok = col < N;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; q++) acc[q] = 0.f;
            float b[CH], bn[CH];
#pragma unroll
            for (int j = 0; j < CH; j++) { const int k = 2 * j + kh; b[j] = (col_ok && k < Kt) ? wrow[k] : 0.f; }
            for (int s0 = 0; s0 < steps; s0 += CH) {
#pragma unroll
                for (int j = 0; j < CH; j++) { const int k = 2 * (s0 + CH + j) + kh; bn[j] = (col_ok && k < Kt && s0 + CH < steps) ? wrow[k] : 0.f; }   // the next chunk's weights do not depend on the activations
#pragma unroll
                for (int j = 0; j < CH; j++) {
                    const int k = 2 * (s0 + j) + kh;
                    const float a = arow[k];                      // (the LDS row is zero beyond the layer's inputs up to its padded width)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[j], acc, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < CH; j++) b[j] = bn[j];
            }
            const float bias = col_ok ? net.bias[i][col] : 0.f;
#pragma unroll
            for (int r = 0; r < (R <= 8 ? 4 : 8); r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row >= R) continue;
                float v = acc[r] + bias;
                if (last) { if (col < LOGIT_LD) f32_row<R, NP>(out, row, LOGIT_LD)[col] = v; }
                else f32_row<R, NP>(out, row, ld)[col] = col_ok ? fmaxf(v, 0.f) : 0.f;
            }
        }
        wave_fence();
        const F32Buf t = in; in = out; out = t;
    }
    const float* zs[R]; int rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int rr = r < n_rows ? r : n_rows - 1; zs[r] = f32_row<R, NP>(in, rr, LOGIT_LD); rows[r] = row0 + rr; }
    const float* const (&zc)[R] = reinterpret_cast<const float* const (&)[R]>(zs);
    policy_head_batched<R>(zc, rows, lane, head, picked);
}

}  // namespace rlinfer
