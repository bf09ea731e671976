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

// Cross-lane steps of the head.  A row lives in HALF a wavefront (32 lanes x 4 columns, n_actions <= 128), so a reduction is five exchanges: four of them
// DPP modifiers on the VALU instruction itself (quad_perm 1032, quad_perm 2301, row_half_mirror, row_mirror: after each, twice as many neighbouring lanes
// hold the same value) and one ds_bpermute (lanes 16 apart).  Round 5 had 64 lanes x 2 columns and six ds_bpermute round trips per reduction.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false)); }
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140;

// NR rows at once, two per pass (lanes 0-31 own row 2p, lanes 32-63 row 2p + 1; an odd last row is done by both halves: same values, same stores).  The
// passes are independent and interleaved, so their cross-lane round trips overlap.  Lane l of a half owns columns l, l + 32, l + 64, l + 96 and draws
// the exponential variates of its columns from ONE philox4 block (stream = row, counter = (call, l)).  Per column the arithmetic is what it always
// was; per row the sampler's stream is not round 5's (then: lane = column mod 64, two words of a block used).
template <int NR, int NCOL>
__device__ __forceinline__ void policy_head_rows_cols(const float* const (&z)[NR], const int (&row)[NR], int lane, const HeadArgs& h, int* picked) {
    // this code is compiled into two translation units with different -ffp-contract settings and must round identically in both (the
    // fused collection kernel promises the batched kernels' bits): pin the setting here
#pragma clang fp contract(fast)
    constexpr int NP = (NR + 1) / 2;
    const int A = h.A; const float inv_temp = h.inv_temp; const int deterministic = h.deterministic; const float* noise = h.noise;
    int32_t* actions = h.actions; float* logp = h.logp; float* probs_out = h.probs_out;
    const int l = lane & 31; const bool hi = lane >= 32;
    bool in[NCOL];
#pragma unroll
    for (int c = 0; c < NCOL; c++) in[c] = (l + 32 * c) < A;
    const float* zp[NP]; int rw[NP];
    float v[NP][NCOL], e[NP][NCOL], pr[NP][NCOL], mx[NP], sum[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        const int n1 = (2 * p + 1 < NR) ? 2 * p + 1 : NR - 1;
        zp[p] = hi ? z[n1] : z[2 * p]; rw[p] = hi ? row[n1] : row[2 * p];
#pragma unroll
        for (int c = 0; c < NCOL; c++) v[p][c] = in[c] ? zp[p][l + 32 * c] * inv_temp : -INFINITY;
        mx[p] = v[p][0];
#pragma unroll
        for (int c = 1; c < NCOL; c++) mx[p] = fmaxf(mx[p], v[p][c]);
    }
#pragma unroll
    for (int p = 0; p < NP; p++) mx[p] = fmaxf(mx[p], dpp_f<DPP_QUAD_1032>(mx[p]));
#pragma unroll
    for (int p = 0; p < NP; p++) mx[p] = fmaxf(mx[p], dpp_f<DPP_QUAD_2301>(mx[p]));
#pragma unroll
    for (int p = 0; p < NP; p++) mx[p] = fmaxf(mx[p], dpp_f<DPP_ROW_HALF_MIRROR>(mx[p]));
#pragma unroll
    for (int p = 0; p < NP; p++) mx[p] = fmaxf(mx[p], dpp_f<DPP_ROW_MIRROR>(mx[p]));
#pragma unroll
    for (int p = 0; p < NP; p++) mx[p] = fmaxf(mx[p], __shfl_xor(mx[p], 16, 64));
#pragma unroll
    for (int p = 0; p < NP; p++) {
#pragma unroll
        for (int c = 0; c < NCOL; c++) e[p][c] = in[c] ? expf(v[p][c] - mx[p]) : 0.f;
        sum[p] = e[p][0];
#pragma unroll
        for (int c = 1; c < NCOL; c++) sum[p] += e[p][c];
    }
#pragma unroll
    for (int p = 0; p < NP; p++) sum[p] += dpp_f<DPP_QUAD_1032>(sum[p]);
#pragma unroll
    for (int p = 0; p < NP; p++) sum[p] += dpp_f<DPP_QUAD_2301>(sum[p]);
#pragma unroll
    for (int p = 0; p < NP; p++) sum[p] += dpp_f<DPP_ROW_HALF_MIRROR>(sum[p]);
#pragma unroll
    for (int p = 0; p < NP; p++) sum[p] += dpp_f<DPP_ROW_MIRROR>(sum[p]);
#pragma unroll
    for (int p = 0; p < NP; p++) sum[p] += __shfl_xor(sum[p], 16, 64);
#pragma unroll
    for (int p = 0; p < NP; p++) {
#pragma unroll
        for (int c = 0; c < NCOL; c++) {
            pr[p][c] = fminf(fmaxf(e[p][c] / sum[p], 1e-11f), 1.f);
            if (probs_out && in[c]) probs_out[(size_t)rw[p] * A + l + 32 * c] = pr[p][c];
        }
    }
    if (!actions) return;
    float best[NP]; int bi[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
        float s[NCOL];
        if (deterministic) {
#pragma unroll
            for (int c = 0; c < NCOL; c++) s[c] = pr[p][c];
        } else {
            if (noise) {   // a recorded tape of variates (the parity tests): the reference's own expression, p / q
#pragma unroll
                for (int c = 0; c < NCOL; c++) s[c] = pr[p][c] / (in[c] ? noise[(size_t)rw[p] * A + l + 32 * c] : 1.f);
            } else {
                uint32_t r[4];
                rlg::philox4(h.seed_lo, h.seed_hi, (uint32_t)rw[p], h.call_ctr, (uint32_t)l, r);
                // q ~ Exp(1): -log(1 - u), u in [0,1) (1 - u is exact).  The hardware logarithm and reciprocal (1 ulp) are all a VARIATE and a score that
                // only orders the columns need; the softmax and the log-probability keep the correctly rounded library functions
#pragma unroll
                for (int c = 0; c < NCOL; c++) {
                    const float q = fmaxf(-0.6931471805599453f * __builtin_amdgcn_logf(1.f - rlg::u32_to_unit(r[c])), 1e-30f);
                    s[c] = pr[p][c] * __builtin_amdgcn_rcpf(q);
                }
            }
        }
        // argmax with lowest-index tie break (torch.argmax / max semantics): ascending columns, strict comparison
        best[p] = in[0] ? s[0] : -INFINITY; bi[p] = l;
#pragma unroll
        for (int c = 1; c < NCOL; c++) { const float sc = in[c] ? s[c] : -INFINITY; if (sc > best[p]) { best[p] = sc; bi[p] = l + 32 * c; } }
    }
#define RLINFER_ARGMAX_STEP(OB, OI)                                                                                               \
    _Pragma("unroll") for (int p = 0; p < NP; p++) {                                                                              \
        const float ob = (OB); const int oi = (OI);                                                                               \
        if (ob > best[p] || (ob == best[p] && oi < bi[p])) { best[p] = ob; bi[p] = oi; }                                          \
    }
    RLINFER_ARGMAX_STEP(dpp_f<DPP_QUAD_1032>(best[p]), dpp_i<DPP_QUAD_1032>(bi[p]))
    RLINFER_ARGMAX_STEP(dpp_f<DPP_QUAD_2301>(best[p]), dpp_i<DPP_QUAD_2301>(bi[p]))
    RLINFER_ARGMAX_STEP(dpp_f<DPP_ROW_HALF_MIRROR>(best[p]), dpp_i<DPP_ROW_HALF_MIRROR>(bi[p]))
    RLINFER_ARGMAX_STEP(dpp_f<DPP_ROW_MIRROR>(best[p]), dpp_i<DPP_ROW_MIRROR>(bi[p]))
    RLINFER_ARGMAX_STEP(__shfl_xor(best[p], 16, 64), __shfl_xor(bi[p], 16, 64))
#undef RLINFER_ARGMAX_STEP
#pragma unroll
    for (int p = 0; p < NP; p++) {
        // the probability of the picked column sits in lane (bi & 31) of this half, in its column slot bi >> 5
        const int cs = bi[p] >> 5;
        float mine = pr[p][0];
#pragma unroll
        for (int c = 1; c < NCOL; c++) mine = cs == c ? pr[p][c] : mine;
        const float pa = __shfl(mine, (lane & 32) | (bi[p] & 31), 64);
        if (l == 0) { actions[rw[p]] = bi[p]; logp[rw[p]] = deterministic ? 0.f : logf(pa); }
        if (picked) {   // uniform over the wavefront
            picked[2 * p] = __builtin_amdgcn_readlane(bi[p], 0);
            if (2 * p + 1 < NR) picked[2 * p + 1] = __builtin_amdgcn_readlane(bi[p], 32);
        }
    }
}
template <int NR>
__device__ __forceinline__ void policy_head_rows(const float* const (&z)[NR], const int (&row)[NR], int lane, const HeadArgs& h, int* picked = nullptr) {
    policy_head_rows_cols<NR, 4>(z, row, lane, h, picked);
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
// What the fused collection kernels are given (one device copy per env batch, written on the env's stream before every collection launch): read through a
// pointer into the constant address space so that the layer table costs scalar loads and no registers between uses
#define RLINFER_CONST __attribute__((address_space(4)))
struct InferPack { InferNet net; HeadArgs head; };
// a wave-uniform pointer that reached a real call's callee in vector registers, made scalar again
template <class T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<T*>(((uint64_t)hi << 32) | (uint64_t)lo);
}
__device__ __forceinline__ HeadArgs head_from_const(const RLINFER_CONST HeadArgs& c) {
    HeadArgs h; h.A = c.A; h.inv_temp = c.inv_temp; h.deterministic = c.deterministic; h.noise = c.noise; h.seed_lo = c.seed_lo; h.seed_hi = c.seed_hi; h.call_ctr = c.call_ctr;
    h.actions = c.actions; h.logp = c.logp; h.probs_out = c.probs_out;
    return h;
}
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

// One 32-column block of a layer with NK K steps, guard-free (see wave_infer): the MFMA chain over the block's weight fragments.  The A operands
// (the wavefront's activation rows) are the same for every column block of a layer: the caller reads them from LDS once per layer.
template <int NK>
__device__ __forceinline__ void mma_block_regs(const bf16x8 (&a)[16], const bf16x8 (&b)[16], f32x16& acc) {
#pragma unroll
    for (int j = 0; j < NK; j++) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[j], acc, 0, 0, 0);
}
// Two column blocks of a layer at once: the same two chains (each block's sum is formed in the order mma_block_regs forms it), issued alternately
template <int NK>
__device__ __forceinline__ void mma_pair_regs(const bf16x8 (&a)[16], const bf16x8 (&b0)[16], const bf16x8 (&b1)[16], f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int j = 0; j < NK; j++) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b0[j], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b1[j], acc1, 0, 0, 0);
    }
}
__device__ __forceinline__ void mma_pair_regs_any(const bf16x8 (&a)[16], const bf16x8 (&b0)[16], const bf16x8 (&b1)[16], f32x16& acc0, f32x16& acc1, int nk) {
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < nk) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b0[j], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b1[j], acc1, 0, 0, 0);
        }
}
// weight fragments of one block, as k_mlp_infer (rlgpu_learn.hip) asks for them
template <int NK>
__device__ __forceinline__ void fetch_block(bf16x8 (&b)[16], const short* w) {
#pragma unroll
    for (int j = 0; j < NK; j++) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)j * 512);
}
__device__ __forceinline__ void fetch_block_any(bf16x8 (&b)[16], const short* w, int nk) {
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < nk) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)j * 512);
}
// The same chain with the A operand read from LDS four steps ahead into four named registers (k_mlp_infer, rlgpu_learn.hip: a wavefront's rows change
// with every tile there).
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
__device__ __forceinline__ void mma_block_any(const short* arow, const bf16x8 (&b)[16], f32x16& acc, int nk) {
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < nk) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + j * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc, 0, 0, 0);
        }
}
// any other depth (<= 16 steps): guarded, slower
__device__ __forceinline__ void mma_block_regs_any(const bf16x8 (&a)[16], const bf16x8 (&b)[16], f32x16& acc, int nk) {
#pragma unroll
    for (int j = 0; j < 16; j++)
        if (j < nk) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[j], acc, 0, 0, 0);
}
// the K depths that occur: 16 (256-wide hidden layers), and the padded observation widths 96 / 128 / 192 / 224 (89, 127, 165, 203 floats)
#define RLINFER_DISPATCH_NK(nk, CALL, FALLBACK) \
    switch (nk) { case 16: CALL(16); break; case 6: CALL(6); break; case 8: CALL(8); break; case 12: CALL(12); break; case 14: CALL(14); break; default: FALLBACK; }

// Policy forward + head for rows row0 .. row0 + R - 1 (R <= 16) whose fp32 observations are obs[r * D + c], by one wavefront.
// Same operand values, accumulation order, bias / ReLU / bf16 rounding and head code as k_mlp_infer (rlgpu_learn.hip): the logits and
// the sampled actions are those of a batched call.  Only the first n_rows (>= 1) of the R rows exist: the others redo the
// last real one (same values, same stores).  buf0 / buf1: LDS, wave_buf_bytes(R, net.ld) each.  picked[r] = the action.
// `NET` is InferNet behind whatever kind of reference the caller has: the collection kernels read it through a pointer into the CONSTANT address space
// (RLINFER_CONST: every field is a scalar load, every loop over layers and blocks a scalar branch), rlgpu_learn.hip's kernels have their own.
template <int R, class NET>
__device__ __forceinline__ void wave_infer(const NET& net, const HeadArgs& head, const float* obs, int row0, int n_rows, short* buf0, short* buf1, int lane, int (&picked)[R],
                                           unsigned long long* prof_split = nullptr) {
    static_assert(R <= WAVE_ROWS, "a wavefront infers at most 16 rows");
    constexpr int CHUNK = 16;
    short* in = buf0; short* out = buf1;
    const int ld = net.ld;
    // A layer's parameters are read when the stream ENTERS the layer, not per column block: `net` sits in the kernel-argument segment, and a field
    // behind a run-time layer index is a scalar load plus a full s_waitcnt wherever it is used -- half a dozen per block, ~1 K cycles of a block
    // that computes for ~0.5 K (tools/probes/infer_probe.hip).  Two sets: the layer whose chain runs (`cur`) and the layer whose weights are
    // being requested (`fet`, up to RLINFER_BUFS - 1 blocks ahead).
    struct LayerP { const short* W; const float* bias; int nk, N, nblk; };
    const int n_layers = net.n_layers;
    auto layer_load = [&](int i) { LayerP q; q.W = net.W[i]; q.bias = net.bias[i]; q.nk = net.K[i] / 16; q.N = net.N[i]; q.nblk = (i == n_layers - 1) ? (net.N[i] + 31) / 32 : net.Npad[i] / 32; return q; };
    LayerP cur = layer_load(0), fet = cur;
    // The weights of a 32-column block are 16 (K = 256) or fewer 1 KB fragments straight from L2, and nothing about them depends on the
    // activations: the NEXT block's are requested before the current block's MFMA chain starts.  For that to overlap anything the wait in
    // front of the chain must be "all but the youngest N loads" (s_waitcnt vmcnt(N)), and the compiler only emits that when N is the same
    // number on every path into the chain -- with a block-dependent number of loads (or a register copy of the prefetched fragments at the top
    // of a loop) it fell back to vmcnt(0), i.e. every block first waited for the block AFTER it (3 K cycles per block for a 512-cycle chain:
    // 80 K cycles per inference; profiles/r05*_collect*.txt).  So: every block issues exactly 16 fragment loads + 1 bias load for its successor
    // (steps beyond the successor's depth re-read its last fragment: same cache line, never used; after the last block the block itself again),
    // into the buffer the chain is NOT reading -- two buffers with fixed roles per half of a loop unrolled by two, no copies.
    int fi = 0, fcb = 0;          // the block whose weights are requested next (RLINFER_BUFS - 1 blocks ahead of the chain)
    auto load_next = [&](bf16x8 (&b)[CHUNK], float& bias) {
        const int nk = fet.nk, cb = fcb;
        const short* w = fet.W + ((size_t)cb * nk * 64 + lane) * 8;
#ifdef PROBE_NO_LOAD
#pragma unroll
        for (int j = 0; j < CHUNK; j++) b[j] = bf16x8{(short)j, 1, 2, 3, 4, 5, 6, (short)cb};
#else
        if (nk == CHUNK) {   // the 256-wide layers: constant offsets from one base (four address adds instead of sixteen clamped ones)
#pragma unroll
            for (int j = 0; j < CHUNK; j++) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)j * 512);
        } else {
#pragma unroll
            for (int j = 0; j < CHUNK; j++) b[j] = *reinterpret_cast<const bf16x8*>(w + (size_t)(j < nk ? j : nk - 1) * 512);
        }
#endif
        const int col = cb * 32 + (lane & 31), nn = fet.N;
        const float bv = fet.bias[col < nn ? col : nn - 1];    // (always issued: a load under a branch would make the count path-dependent)
        bias = col < nn ? bv : 0.f;
        // the block after it; the last block's successor is the last block itself (its loads keep the count, nothing uses them)
        if (fcb + 1 < fet.nblk) fcb++;
        else if (fi + 1 < n_layers) { fi++; fcb = 0; fet = layer_load(fi); }
    };
    int li = 0, lcb = 0;          // the block whose MFMA chain runs next
    bf16x8 areg[CHUNK];           // the current layer's A operands: this lane's 16-byte slice of its tile row, one per K step
    auto run_block = [&](const bf16x8 (&b)[CHUNK], float bias, bf16x8 (&bn)[CHUNK], float& bias_n) {
        const int i = li, cb = lcb;
        const bool last = (i == n_layers - 1);
        const int N = cur.N, nk = cur.nk, nblk = cur.nblk;
        load_next(bn, bias_n);
        float* const logits = reinterpret_cast<float*>(out);
        if (cb == 0) {
            // tile rows >= R do not exist: their lanes re-read row 0 (rows of an MFMA are independent and only rows < R are stored), which
            // keeps the operand loads free of exec masking
            const short* arow = in + ((lane & 31) < R ? (lane & 31) : 0) * ld + 8 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < CHUNK; j++) areg[j] = *reinterpret_cast<const bf16x8*>(arow + (j < nk ? j : 0) * 16);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
#ifdef PROBE_NO_MMA
        for (int j = 0; j < 16; j++) acc[j] = (float)b[j][0] + (float)areg[j][0];
#else
#define RLINFER_MMA(NK) mma_block_regs<NK>(areg, b, acc)
        RLINFER_DISPATCH_NK(nk, RLINFER_MMA, mma_block_regs_any(areg, b, acc, nk))
#undef RLINFER_MMA
#endif
        // C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5): rows 0..7 are registers 0..3, rows 8..15 registers 4..7
        const int col = cb * 32 + (lane & 31);
        if (last) {
#pragma unroll
            for (int r = 0; r < (R <= 8 ? 4 : 8); r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) logits[row * LOGIT_LD + col] = acc[r] + bias;
            }
        } else {
#pragma unroll
            for (int r = 0; r < (R <= 8 ? 4 : 8); r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) out[row * ld + col] = (col < N) ? to_bf16(fmaxf(acc[r] + bias, 0.f)) : (short)0;
            }
        }
        const bool layer_done = cb + 1 >= nblk;
        if (layer_done) { wave_fence(); short* t = in; in = out; out = t; }   // the layer is complete: its outputs are the next layer's inputs
#ifdef RLG_TICK_PROFILE
        if (layer_done && prof_split && i < 6) prof_split[2 + i] = __builtin_amdgcn_s_memtime();
#endif
        if (layer_done) { li = i + 1; lcb = 0; if (li < n_layers) cur = layer_load(li); } else lcb = cb + 1;
    };
#ifndef RLINFER_PAIR
#define RLINFER_PAIR 1   /* column blocks per pass of the block loop: 2 = a pair shares the A operands, the bookkeeping and the output stage's control flow (round 6) */
#endif
#if RLINFER_PAIR
    // Two column blocks per pass.  What a block costs is ~1.5 K cycles of scalar / address / control instructions around a 0.5 K chain (one wavefront issuing
    // alone); a pair pays them once.  Blocks cb and cb + 1 (an odd last block is paired with itself: same values, same stores) -- four fragment buffers, two being
    // read by the chains while the next pair's two are on their way; every pass issues exactly 32 fragment loads + 2 bias loads (see above).
    auto load_next2 = [&](bf16x8 (&b0)[CHUNK], float& bias_0, bf16x8 (&b1)[CHUNK], float& bias_1) {
        const int nk = fet.nk, cb0 = fcb, cb1 = (fcb + 1 < fet.nblk) ? fcb + 1 : fcb;
        const short* w0 = fet.W + ((size_t)cb0 * nk * 64 + lane) * 8;
        const short* w1 = fet.W + ((size_t)cb1 * nk * 64 + lane) * 8;
        if (nk == CHUNK) {
#pragma unroll
            for (int j = 0; j < CHUNK; j++) { b0[j] = *reinterpret_cast<const bf16x8*>(w0 + (size_t)j * 512); b1[j] = *reinterpret_cast<const bf16x8*>(w1 + (size_t)j * 512); }
        } else {
#pragma unroll
            for (int j = 0; j < CHUNK; j++) {
                const size_t o = (size_t)(j < nk ? j : nk - 1) * 512;
                b0[j] = *reinterpret_cast<const bf16x8*>(w0 + o); b1[j] = *reinterpret_cast<const bf16x8*>(w1 + o);
            }
        }
        const int nn = fet.N, c0 = cb0 * 32 + (lane & 31), c1 = cb1 * 32 + (lane & 31);
        const float bv0 = fet.bias[c0 < nn ? c0 : nn - 1], bv1 = fet.bias[c1 < nn ? c1 : nn - 1];
        bias_0 = c0 < nn ? bv0 : 0.f; bias_1 = c1 < nn ? bv1 : 0.f;
        if (fcb + 2 < fet.nblk) fcb += 2;
        else if (fi + 1 < n_layers) { fi++; fcb = 0; fet = layer_load(fi); }
    };
    auto run_pair = [&](const bf16x8 (&b0)[CHUNK], float bias_0, const bf16x8 (&b1)[CHUNK], float bias_1,
                        bf16x8 (&bn0)[CHUNK], float& bias_n0, bf16x8 (&bn1)[CHUNK], float& bias_n1) {
        const int i = li, cb = lcb;
        const bool last = (i == n_layers - 1);
        const int N = cur.N, nk = cur.nk, nblk = cur.nblk;
        const int cb1 = (cb + 1 < nblk) ? cb + 1 : cb;
        load_next2(bn0, bias_n0, bn1, bias_n1);
        float* const logits = reinterpret_cast<float*>(out);
        if (cb == 0) {
            const short* arow = in + ((lane & 31) < R ? (lane & 31) : 0) * ld + 8 * (lane >> 5);
#pragma unroll
            for (int j = 0; j < CHUNK; j++) areg[j] = *reinterpret_cast<const bf16x8*>(arow + (j < nk ? j : 0) * 16);
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; r++) { acc0[r] = 0.f; acc1[r] = 0.f; }
#define RLINFER_MMA2(NK) mma_pair_regs<NK>(areg, b0, b1, acc0, acc1)
        RLINFER_DISPATCH_NK(nk, RLINFER_MMA2, mma_pair_regs_any(areg, b0, b1, acc0, acc1, nk))
#undef RLINFER_MMA2
        const int col0 = cb * 32 + (lane & 31), col1 = cb1 * 32 + (lane & 31);
        if (last) {
#pragma unroll
            for (int r = 0; r < (R <= 8 ? 4 : 8); r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) { logits[row * LOGIT_LD + col0] = acc0[r] + bias_0; logits[row * LOGIT_LD + col1] = acc1[r] + bias_1; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < (R <= 8 ? 4 : 8); r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < R) {
                    out[row * ld + col0] = (col0 < N) ? to_bf16(fmaxf(acc0[r] + bias_0, 0.f)) : (short)0;
                    out[row * ld + col1] = (col1 < N) ? to_bf16(fmaxf(acc1[r] + bias_1, 0.f)) : (short)0;
                }
            }
        }
        const bool layer_done = cb + 2 >= nblk;
        if (layer_done) { wave_fence(); short* t = in; in = out; out = t; }
#ifdef RLG_TICK_PROFILE
        if (layer_done && prof_split && i < 6) prof_split[2 + i] = __builtin_amdgcn_s_memtime();
#endif
        if (layer_done) { li = i + 1; lcb = 0; if (li < n_layers) cur = layer_load(li); } else lcb = cb + 2;
    };
    {
        bf16x8 B0a[CHUNK], B0b[CHUNK], B1a[CHUNK], B1b[CHUNK]; float bias0a, bias0b, bias1a = 0.f, bias1b = 0.f;
        load_next2(B0a, bias0a, B0b, bias0b);
        // (the first blocks' weights are on their way while the observations are converted: they do not depend on them)
        {   // every value of the R rows is asked for before the first one is used: a load inside `if (c < D)` inside a loop waits for itself (16 serial L2
            // round trips for 8 rows of 89 floats: 11 K cycles).  Lanes past the row's end read its last element and store a zero.
            const int D = net.D, K0 = net.K[0];
            constexpr int CH = 4;                 // 64-column chunks per row held in registers at once (K0 <= 256)
            for (int c0 = 0; c0 < K0; c0 += 64 * CH) {
                float v[R][CH];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int rr = r < n_rows ? r : n_rows - 1;
#pragma unroll
                    for (int q = 0; q < CH; q++) { const int c = c0 + 64 * q + lane; v[r][q] = obs[(size_t)rr * D + (c < D ? c : D - 1)]; }
                }
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int q = 0; q < CH; q++) { const int c = c0 + 64 * q + lane; if (c < K0) in[r * ld + c] = to_bf16(c < D ? v[r][q] : 0.f); }
            }
        }
        wave_fence();
#ifdef RLG_TICK_PROFILE
        if (prof_split) prof_split[1] = __builtin_amdgcn_s_memtime();   // profiler build: observations staged
#endif
        while (true) {
            run_pair(B0a, bias0a, B0b, bias0b, B1a, bias1a, B1b, bias1b); if (li >= n_layers) break;
            run_pair(B1a, bias1a, B1b, bias1b, B0a, bias0a, B0b, bias0b); if (li >= n_layers) break;
        }
    }
#else
    {
#ifndef RLINFER_BUFS
#define RLINFER_BUFS 2   /* weight buffers: the chain reads one while RLINFER_BUFS - 1 blocks are on their way (measured with tools/probes/infer_probe.hip: 2 and 3 run alike -- the loads are 5 K of an inference's cycles) */
#endif
        bf16x8 B0[CHUNK], B1[CHUNK]; float bias0, bias1 = 0.f;
        load_next(B0, bias0);
#if RLINFER_BUFS >= 3
        bf16x8 B2[CHUNK]; float bias2 = 0.f;
        load_next(B1, bias1);
#endif
#if RLINFER_BUFS >= 4
        bf16x8 B3[CHUNK]; float bias3 = 0.f;
        load_next(B2, bias2);
#endif
        // (the first blocks' weights are on their way while the observations are converted: they do not depend on them)
        {   // every value of the R rows is asked for before the first one is used: a load inside `if (c < D)` inside a loop waits for itself (16 serial L2
            // round trips for 8 rows of 89 floats: 11 K cycles).  Lanes past the row's end read its last element and store a zero.
            const int D = net.D, K0 = net.K[0];
            constexpr int CH = 4;                 // 64-column chunks per row held in registers at once (K0 <= 256)
            for (int c0 = 0; c0 < K0; c0 += 64 * CH) {
                float v[R][CH];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const int rr = r < n_rows ? r : n_rows - 1;
#pragma unroll
                    for (int q = 0; q < CH; q++) { const int c = c0 + 64 * q + lane; v[r][q] = obs[(size_t)rr * D + (c < D ? c : D - 1)]; }
                }
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int q = 0; q < CH; q++) { const int c = c0 + 64 * q + lane; if (c < K0) in[r * ld + c] = to_bf16(c < D ? v[r][q] : 0.f); }
            }
        }
        wave_fence();
#ifdef RLG_TICK_PROFILE
        if (prof_split) prof_split[1] = __builtin_amdgcn_s_memtime();   // profiler build: observations staged
#endif
        while (true) {
#if RLINFER_BUFS == 2
            run_block(B0, bias0, B1, bias1); if (li >= n_layers) break;
            run_block(B1, bias1, B0, bias0); if (li >= n_layers) break;
#elif RLINFER_BUFS == 3
            run_block(B0, bias0, B2, bias2); if (li >= n_layers) break;
            run_block(B1, bias1, B0, bias0); if (li >= n_layers) break;
            run_block(B2, bias2, B1, bias1); if (li >= n_layers) break;
#else
            run_block(B0, bias0, B3, bias3); if (li >= n_layers) break;
            run_block(B1, bias1, B0, bias0); if (li >= n_layers) break;
            run_block(B2, bias2, B1, bias1); if (li >= n_layers) break;
            run_block(B3, bias3, B2, bias2); if (li >= n_layers) break;
#endif
        }
    }
#endif
#ifdef RLG_TICK_PROFILE
    if (prof_split) *prof_split = __builtin_amdgcn_s_memtime();   // profiler build: the MLP ends here, the head begins
#endif
    const float* logits = reinterpret_cast<const float*>(in);
    const float* zs[R]; int rows[R];
#pragma unroll
    for (int r = 0; r < R; r++) { const int rr = r < n_rows ? r : n_rows - 1; zs[r] = logits + rr * LOGIT_LD; rows[r] = row0 + rr; }
    const float* const (&zc)[R] = reinterpret_cast<const float* const (&)[R]>(zs);
#ifdef PROBE_NO_HEAD
    for (int r = 0; r < R; r++) picked[r] = (int)zc[r][0];
#else
    policy_head_batched<R>(zc, rows, lane, head, picked);
#endif
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

template <int R, int NP, class NET>
__device__ __forceinline__ void wave_infer_f32(const NET& net, const HeadArgs& head, const float* obs, int row0, int n_rows, F32Buf in, F32Buf out, int lane, int (&picked)[R]) {
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
