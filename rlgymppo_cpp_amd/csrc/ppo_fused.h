// ppo_fused.h -- one PPO minibatch of the flagship networks (obs -> 256 x 3 -> 90 / 1, bf16 operands, fp32 sums) in TWO kernels.
//
//   k_ppo_fwd_bwd   per 128-row stripe and network, everything between the experience rows and the activation gradients in ONE launch:
//                   gather through the shuffled index list + bf16 staging (was k_rows_to_bf16), the four forward layers (were 4 k_gemm_nt
//                   launches per network), the PPO policy loss / the value loss with their metrics (were k_ppo_policy_loss / k_value_loss,
//                   PPOLearner.cpp:148-198, DiscretePolicy.cpp:64-75), and the three input-gradient layers with the ReLU masks (were 3 k_gemm_nt
//                   launches per network).  Activations travel from layer to layer in LDS, the ReLU masks stay in registers (the wavefront that
//                   produced a tile of a hidden layer masks the same tile of its gradient), the logits never leave LDS.
//   k_dw_grouped    dW_i += dZ_i^T . A_(i-1), db_i += column sums of dZ_i for every layer of both networks in one launch (was 8 k_gemm_tn
//                   launches on two side streams): a workgroup owns one layer's WHOLE dW for a slab of rows, so each operand byte is read once
//                   (the 128 x 128 tiles read each twice), the row slabs go through a three-deep register ring into double-buffered LDS tiles and
//                   from there through ds_read_b64_tr_b16 into the MFMAs; the bias gradient is one extra MFMA column against a register of ones.
//
// What still crosses HBM between the two: the hidden activations and the activation gradients (bf16, written once, read once) -- the dW sum runs
// over ALL rows of the minibatch, and a CU's registers hold a quarter of one network's dW, so it cannot be folded into the stripe.
// MFMA operand order: the WEIGHT fragment is the A operand and the activation rows the B operand (out^T = W . in^T), so a lane ends up with four
// runs of four consecutive features of ONE row -- 8-byte LDS stores instead of sixteen 2-byte ones.  Rounding points are the per-layer path's:
// bf16 hidden activations, bf16 activation gradients, fp32 logits / values / dW.
#pragma once

namespace fused {

constexpr int R = 128;           // rows per stripe (workgroup)
constexpr int RT = R / 32;       // 32-row MFMA tiles per stripe
constexpr int NWAVE = 8;         // 512 threads: wavefront w owns the 32-feature block w of every hidden layer
constexpr int H = 256;           // hidden width this kernel is built for
constexpr int LD = 264;          // LDS activation row, bf16 elements (528 B: the 16 rows of a ds_read_b128 group land on distinct bank quads)
constexpr int LDF = 100;         // LDS logits row, floats (400 B: 16 consecutive rows start on distinct bank quads)
constexpr int BIAS_FLOATS = 3 * H + 128;   // the three hidden layers' biases and the last layer's (zero padded), staged in LDS behind the activation buffers
constexpr size_t SMEM_BYTES = (size_t)2 * R * LD * sizeof(short) + BIAS_FLOATS * sizeof(float);

struct NetArgs {
    const short* wf[4];          // forward fragments  (k_weight_frags of W   [N pad][kp in])
    const short* wtf[4];         // backward fragments (k_weight_frags of W^T [K pad][kp out]); [0] unused
    const float* bias[4];
    short* act[3];               // hidden activations [rows][256] bf16 (for k_dw_grouped)
    short* dy[4];                // dL/d(pre-activation) of layer i, [rows][kp out] bf16 (for k_dw_grouped)
    int out_dim;                 // 90 (policy: n_actions) / 1 (critic)
};
struct Args {
    const float* obs; const int32_t* idx; int rows; int D;
    int row0;                    // the launch covers rows [row0, rows) (chunked launches)
    short* x16;                  // [rows][K0P] bf16: the gathered input (k_dw_grouped's operand for layer 0)
    NetArgs net[2];              // policy, critic
    const int32_t* actions; const float* old_logp; const float* adv; const float* targets;
    float inv_temp, clip, ent_coef, scale;   // scale = ratio / rows
    int debug;                   // experiments only (FZ_DEBUG builds): 1 no copies to HBM, 2 no MFMAs, 4 no epilogues
    unsigned long long* prof;    // profiling builds of the host only (RLGPU_FUSED_PROF): cycles per phase summed over the workgroups' first wavefronts
    float loss_scale;            // the factor the loss gradient is multiplied by when loss_scale_dev is null: 1 (bf16 mode)
    const float* loss_scale_dev; // fp16 mode: the dynamic loss scale, kept and updated ON THE DEVICE (rlgpu_learn.hip k_ls_decide) so that an optimizer step
                                 // needs no answer from the host -- with collectionDuringLearn the host must get on to the next collection launch
    float* metrics;              // [0] entropy, [1] KL, [2] clip count, [3] ratio, [4] value squared error: sums over rows
};

using bf16x8 = __attribute__((ext_vector_type(8))) short;
using bf16x4 = __attribute__((ext_vector_type(4))) short;
using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ short f2bf_(float f) { __hip_bfloat16 h = __float2bfloat16(f); return *reinterpret_cast<short*>(&h); }
// Operand type of the kernels below: bf16 (autocastLearn as the reference's FrameworkTorch.h:14 configures it) or, HALF = true, fp16 (BASELINE
// configs[4]'s wording) -- with fp16 the loss gradient is multiplied by a dynamic loss scale (LossScale below).  Both are 16-bit patterns whose
// top bit is the sign, so the packed ReLU / "positive" tricks of the epilogues hold for either.
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
template <bool HALF> __device__ __forceinline__ short f2s(float f) {
    if constexpr (HALF) { const _Float16 h = (_Float16)f; return __builtin_bit_cast(short, h); }
    else return f2bf_(f);
}
template <bool HALF> __device__ __forceinline__ f32x16 mma16(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// (The dynamic loss scale of the fp16 mode -- PRIV/Util/gradscaler.hpp:26-34: init 2^16, growth 2 every 2000 clean steps, backoff 0.5 -- is host
// state of the learner, rlgpu_learn.hip rlgpu_clip_adam_step; the kernels get it as a number.)
__device__ __forceinline__ float bf2f_(short s) { return __uint_as_float(((unsigned int)(unsigned short)s) << 16); }

template <int NK> struct Fr { bf16x8 f[NK]; };
template <int NK>
__device__ __forceinline__ void load_fr(Fr<NK>& F, const short* wf, int cb, int lane) {
    const short* p = wf + ((size_t)cb * NK * 64 + lane) * 8;
#pragma unroll
    for (int s = 0; s < NK; s++) F.f[s] = *reinterpret_cast<const bf16x8*>(p + (size_t)s * 512);
}
// acc[rt][feature m of the block][row] = init + W-fragments . in[32 rt + row][K].  `init` is the first MFMA's C operand (the bias of a forward
// layer, zeros for a gradient: no accumulator is written before the MFMAs); the activation operands of step s + 1 are requested before step
// s's MFMAs are issued.
template <int NK, bool HALF>
__device__ __forceinline__ void mma_tiles(const Fr<NK>& F, const short* in, int lane, const f32x16& init, f32x16 (&acc)[RT]) {
    const short* brow = in + (lane & 31) * LD + 8 * (lane >> 5);
    bf16x8 b[2][RT];
#pragma unroll
    for (int r = 0; r < RT; r++) b[0][r] = *reinterpret_cast<const bf16x8*>(brow + r * 32 * LD);
#pragma unroll
    for (int s = 0; s < NK; s++) {
        if (s + 1 < NK) {
#pragma unroll
            for (int r = 0; r < RT; r++) b[(s + 1) & 1][r] = *reinterpret_cast<const bf16x8*>(brow + r * 32 * LD + (s + 1) * 16);
        }
#pragma unroll
        for (int r = 0; r < RT; r++) acc[r] = mma16<HALF>(F.f[s], b[s & 1][r], s == 0 ? init : acc[r]);
    }
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; q++) z[q] = 0.f;
    return z;
}
// C layout of the 32x32 tile with the weights as the A operand: lane l holds row (l & 31) of the row tile and features
// 8 g + 4 (l >> 5) + j of the block for register 4 g + j
// hidden layer: relu(acc + bias) -> bf16 -> out[row][cb * 32 + ...], mask bit (rt * 16 + q) = activation > 0
// the lane's 16 bias values of feature block cb, in accumulator order, from the LDS copy (stage_bias): a global load here would sit right in
// front of the MFMA that takes them as its C operand
__device__ __forceinline__ f32x16 load_bias(const float* bias_lds, int cb, int lane) {
    f32x16 r;
    const int nb = cb * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const float4 v = *reinterpret_cast<const float4*>(bias_lds + nb + 8 * g);
        r[4 * g] = v.x; r[4 * g + 1] = v.y; r[4 * g + 2] = v.z; r[4 * g + 3] = v.w;
    }
    return r;
}
// all four layers' biases -> LDS (hidden layer i at [i * H, + H), the last layer at [3 H, + 128) zero padded); visible after the next barrier
__device__ __forceinline__ void stage_bias(float* bias_lds, const float* const (&bias)[4], int out_dim, int tid) {
    for (int i = tid; i < BIAS_FLOATS; i += 512) {
        const int layer = i / H < 3 ? i / H : 3, k = i - layer * H;
        bias_lds[i] = (layer < 3 || k < out_dim) ? bias[layer][k] : 0.f;
    }
}
using s16x2 = __attribute__((ext_vector_type(2))) short;
using u16x2 = __attribute__((ext_vector_type(2))) unsigned short;
struct Mask { unsigned int m[2]; };   // ReLU mask of a wavefront's 4 tiles: pair p = 8 r + 2 g + h of (register 4 g + 2 h, + 1) -> bits (p & 15) and 16 + (p & 15) of m[p >> 4]
template <bool HALF>
__device__ __forceinline__ unsigned int pack16(float a, float b) {
    s16x2 v; v[0] = f2s<HALF>(a); v[1] = f2s<HALF>(b);
    return __builtin_bit_cast(unsigned int, v);
}
// hidden layer (accumulators started from the bias): relu -> bf16 -> out[row][cb * 32 + ...].  On packed pairs: max with 0 as SIGNED 16-bit integers
// is the ReLU of a bf16 pair, min with 1 as UNSIGNED ones is "positive" -- two instructions per pair where the fp32 form took six per element.
template <bool HALF>
__device__ __forceinline__ Mask epilogue_hidden(const f32x16 (&acc)[RT], int cb, short* out, int lane) {
    Mask mk; mk.m[0] = 0; mk.m[1] = 0;
    const int nb = cb * 32 + 4 * (lane >> 5);
    const s16x2 zero2 = {0, 0}; const u16x2 one2 = {1, 1};
#pragma unroll
    for (int r = 0; r < RT; r++) {
        short* orow = out + (r * 32 + (lane & 31)) * LD + nb;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned int pk[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const s16x2 v = __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack16<HALF>(acc[r][4 * g + 2 * h], acc[r][4 * g + 2 * h + 1])), zero2);
                pk[h] = __builtin_bit_cast(unsigned int, v);
                const unsigned int t = __builtin_bit_cast(unsigned int, __builtin_elementwise_min(__builtin_bit_cast(u16x2, v), one2));
                const int p = 8 * r + 2 * g + h;
                mk.m[p >> 4] |= t << (p & 15);
            }
            *reinterpret_cast<uint2*>(orow + 8 * g) = make_uint2(pk[0], pk[1]);
        }
    }
    // (pins the mask words HERE: left alone, the compiler sinks their computation to the first use -- the backward pass -- and keeps every
    // activation alive in scratch until then)
    asm volatile("" : "+v"(mk.m[0]), "+v"(mk.m[1]));
    return mk;
}
// input gradient of a hidden layer: bf16(acc) where the forward activation was positive
template <bool HALF>
__device__ __forceinline__ void epilogue_dx(const f32x16 (&acc)[RT], const Mask& mk, int cb, short* out, int lane) {
    const int nb = cb * 32 + 4 * (lane >> 5);
    const u16x2 zero2 = {0, 0};
#pragma unroll
    for (int r = 0; r < RT; r++) {
        short* orow = out + (r * 32 + (lane & 31)) * LD + nb;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned int pk[2];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int p = 8 * r + 2 * g + h;
                const unsigned int t = (mk.m[p >> 4] >> (p & 15)) & 0x00010001u;
                const u16x2 keep = zero2 - __builtin_bit_cast(u16x2, t);          // 0xFFFF where the activation was positive
                pk[h] = pack16<HALF>(acc[r][4 * g + 2 * h], acc[r][4 * g + 2 * h + 1]) & __builtin_bit_cast(unsigned int, keep);
            }
            *reinterpret_cast<uint2*>(orow + 8 * g) = make_uint2(pk[0], pk[1]);
        }
    }
}
// rows [m0, m0 + R) x W columns of an LDS stripe -> global bf16 [rows][W], 16 bytes per lane
template <int W>
__device__ __forceinline__ void copy_out(const short* buf, short* dst, int m0, int rows, int tid) {
    constexpr int CPR = W / 8, N = (R * CPR + 511) / 512;
    // a uniform base + a 32-bit lane offset per piece (one address register each; with 64-bit lane addresses the compiler computed every
    // piece of every layer's copy up front and spilled them)
    char* const base = reinterpret_cast<char*>(dst + (size_t)m0 * W);
    const int nrow = rows - m0;
#pragma unroll
    for (int it = 0; it < N; it++) {
        const int idx = tid + it * 512, row = idx / CPR, ch = idx % CPR;
        const unsigned off = (unsigned)(row * W + ch * 8) * 2u;
        if (idx < R * CPR && row < nrow) *reinterpret_cast<uint4*>(base + off) = *reinterpret_cast<const uint4*>(buf + row * LD + ch * 8);
    }
}
__device__ __forceinline__ float quad_sum(float x) {
    x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); return x;
}
__device__ __forceinline__ float quad_max(float x) {
    x = fmaxf(x, __shfl_xor(x, 1, 64)); x = fmaxf(x, __shfl_xor(x, 2, 64)); return x;
}
__device__ __forceinline__ float wave_sum_(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

#ifdef FZ_DEBUG
#define FZ_DBG(b) (g.debug & (b))
#else
#define FZ_DBG(b) false
#endif
template <int K0P, int OUTP, bool POLICY, bool HALF>
__device__ __forceinline__ void stripe_body(const Args& g, const NetArgs& n, short* buf0, short* buf1, float (*red)[4]) {
    constexpr int NK0 = K0P / 16, NKH = H / 16, NKO = OUTP / 16;
    constexpr int OB = OUTP / 32;                    // 32-feature blocks of the last layer
    constexpr int T3 = (OB * RT + NWAVE - 1) / NWAVE;   // last-layer tiles per wavefront
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = g.row0 + blockIdx.x * R;
#ifdef FZ_PROF   // profiling build (tools/fused_prof.sh, RLGPU_FUSED_PROF=1): cycles per phase of every workgroup's first lane; the stamps stay in
                 // registers and go out in one burst at the end (an atomic per stamp would sit in front of every later load's wait)
    unsigned long long t_st[11]; int n_st = 0;
    t_st[0] = clock64();
    auto stamp = [&](int k) { t_st[k + 1] = clock64(); n_st = k + 1; };
#else
    auto stamp = [](int) {};
#endif

    Fr<NK0> f0; load_fr(f0, n.wf[0], w, lane);       // lands under the gather
    float* const bias_lds = reinterpret_cast<float*>(buf0 + 2 * R * LD);
    stage_bias(bias_lds, n.bias, n.out_dim, tid);
    // the loss's per-row inputs, requested now (policy: row tid >> 2, as the loss phase deals the rows; critic: row tid)
    int pre_a = 0; float pre_olp = 0.f, pre_adv = 0.f, pre_tgt = 0.f;
    {   // gather + bf16 staging: four lanes per row, pieces of 8 columns.  Every load is unconditional, so that all of a lane's loads are in flight
        // together: whole pieces as two 16-byte loads (the rows are only 4-byte aligned: gfx950 global loads take that), the row's last, partial
        // piece as up to 8 clamped scalar loads made by every lane and used by the one that owns the piece
        struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
        const int row = tid >> 2, sub = tid & 3, gm = m0 + row;
        const bool valid = gm < g.rows;
        const int gmc = valid ? gm : g.rows - 1;
        const int srow = g.idx ? g.idx[gmc] : gmc;
        const float* src = g.obs + (size_t)srow * g.D;
        if (POLICY) { pre_a = g.actions[srow]; pre_olp = g.old_logp[srow]; pre_adv = g.adv[srow]; }
        else { const int gt = m0 + (tid & (R - 1)); const int gtc = gt < g.rows ? gt : g.rows - 1; pre_tgt = g.targets[g.idx ? g.idx[gtc] : gtc]; }
        const int full = g.D >> 3;                     // whole pieces; piece `full` holds the D & 7 last columns
        F4 va[K0P / 32], vb[K0P / 32]; float tail[8];
#pragma unroll
        for (int jj = 0; jj < K0P / 32; jj++) {
            const int pi = sub + 4 * jj, pc = pi < full ? pi : full - 1;
            va[jj] = *reinterpret_cast<const F4*>(src + pc * 8); vb[jj] = *reinterpret_cast<const F4*>(src + pc * 8 + 4);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) tail[q] = src[full * 8 + q < g.D ? full * 8 + q : g.D - 1];
#pragma unroll
        for (int jj = 0; jj < K0P / 32; jj++) {
            const int pi = sub + 4 * jj, c0 = pi * 8;
            float x[8] = {va[jj].x, va[jj].y, va[jj].z, va[jj].w, vb[jj].x, vb[jj].y, vb[jj].z, vb[jj].w};
            bf16x8 v;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float e = pi < full ? x[q] : ((pi == full && c0 + q < g.D) ? tail[q] : 0.f);
                v[q] = valid ? f2s<HALF>(e) : (short)0;
            }
            *reinterpret_cast<bf16x8*>(buf0 + row * LD + c0) = v;
            if (POLICY && valid) *reinterpret_cast<bf16x8*>(g.x16 + (size_t)gm * K0P + c0) = v;
        }
    }
    __syncthreads();
    stamp(0);
    f32x16 acc[RT];
    // ---- forward ----  per layer: [barrier] request the NEXT layer's fragments, copy the previous output to HBM, MFMAs, epilogue.  The
    // fragment requests come before the copy's stores: memory operations retire in order, so a store issued first would gate them.
    Fr<NKH> f1, f2, f3, f3b;
    load_fr(f1, n.wf[1], w, lane);
    f32x16 bi = load_bias(bias_lds, w, lane);
    if (!FZ_DBG(2)) mma_tiles<NK0, HALF>(f0, buf0, lane, bi, acc);
    const Mask mask0 = epilogue_hidden<HALF>(acc, w, buf1, lane);
    __syncthreads();
    stamp(1);
    load_fr(f2, n.wf[2], w, lane);
    bi = load_bias(bias_lds + H, w, lane);
    if (!FZ_DBG(1)) copy_out<H>(buf1, n.act[0], m0, g.rows, tid);
    if (!FZ_DBG(2)) mma_tiles<NKH, HALF>(f1, buf1, lane, bi, acc);
    const Mask mask1 = epilogue_hidden<HALF>(acc, w, buf0, lane);
    __syncthreads();
    stamp(2);
    // last-layer tiles t = w, w + 8 (feature block t / RT, row tile t % RT); fragments of a block that does not exist are clamped (loaded, unused)
    load_fr(f3, n.wf[3], (w / RT < OB) ? w / RT : OB - 1, lane);
    bi = load_bias(bias_lds + 2 * H, w, lane);
    if (!FZ_DBG(1)) copy_out<H>(buf0, n.act[1], m0, g.rows, tid);
    if (!FZ_DBG(2)) mma_tiles<NKH, HALF>(f2, buf0, lane, bi, acc);
    if (T3 > 1) load_fr(f3b, n.wf[3], ((w + NWAVE) / RT < OB) ? (w + NWAVE) / RT : OB - 1, lane);
    const Mask mask2 = epilogue_hidden<HALF>(acc, w, buf1, lane);
    __syncthreads();
    stamp(3);
    Fr<NKO> g3; load_fr(g3, n.wtf[3], w, lane);      // lands under the last layer and the loss
    if (!FZ_DBG(1)) copy_out<H>(buf1, n.act[2], m0, g.rows, tid);
    // last layer: fp32 logits / values -> buf0 as float [R][LDF]
    float* const zf = reinterpret_cast<float*>(buf0);
#pragma unroll
    for (int k = 0; k < T3; k++) {
        const int t = w + k * NWAVE;
        const Fr<NKH>& fk = k == 0 ? f3 : f3b;
        if (t < OB * RT) {
            const int cb = t / RT, rt = t % RT;
            f32x16 a1;
#pragma unroll
            for (int q = 0; q < 16; q++) a1[q] = 0.f;
            const short* brow = buf1 + (rt * 32 + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
            for (int s = 0; s < NKH; s++) a1 = mma16<HALF>(fk.f[s], *reinterpret_cast<const bf16x8*>(brow + s * 16), a1);
            const int nb = cb * 32 + 4 * (lane >> 5);
            float* zrow = zf + (rt * 32 + (lane & 31)) * LDF + nb;
#pragma unroll
            for (int gq = 0; gq < 4; gq++) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + 3 * H + nb + 8 * gq);      // (zero beyond the layer's outputs)
                float4 v;
                v.x = a1[4 * gq + 0] + b4.x; v.y = a1[4 * gq + 1] + b4.y; v.z = a1[4 * gq + 2] + b4.z; v.w = a1[4 * gq + 3] + b4.w;
                *reinterpret_cast<float4*>(zrow + 8 * gq) = v;
            }
        }
    }
    __syncthreads();
    stamp(4);
    // ---- loss: d(loss)/d(last layer output) -> buf1 [R][OUTP] bf16 ----
    if (POLICY) {
        constexpr int PER = OUTP / 4;
        const int row = tid >> 2, sub = tid & 3, gm = m0 + row, A = n.out_dim;
        const bool valid = gm < g.rows;
        const float* z = zf + row * LDF + sub;
        // four lanes per row, element j of lane `sub` = logit sub + 4 j.  v_exp_f32 / v_log_f32 and one reciprocal per row (the results leave as bf16)
        float s[PER], lp[PER];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < PER; j++) { s[j] = (sub + 4 * j < A) ? z[4 * j] * g.inv_temp : -INFINITY; mx = fmaxf(mx, s[j]); }
        mx = quad_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PER; j++) { s[j] = __expf(s[j] - mx); sum += s[j]; }     // (exp(-inf) = 0 for the padding)
        sum = quad_sum(sum);
        const float inv_sum = 1.f / sum;
        const int a = pre_a;
        float ent = 0.f, lpa = 0.f, pa = 0.f;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            s[j] = s[j] * inv_sum;
            const float p = fminf(fmaxf(s[j], 1e-11f), 1.f);
            lp[j] = __logf(p);
            if (sub + 4 * j < A) ent -= lp[j] * p;
            if (sub + 4 * j == a) { lpa = lp[j]; pa = p; }
        }
        ent = quad_sum(ent); lpa = quad_sum(lpa); pa = quad_sum(pa);
        const float olp = pre_olp, ad = pre_adv;
        const float ratio = __expf(lpa - olp);
        const float clipped = fminf(fmaxf(ratio, 1.f - g.clip), 1.f + g.clip);
        const float surr1 = ratio * ad, surr2 = clipped * ad;
        // d(-min(surr1, surr2)) / d logp: torch.min splits ties, and inside the clip range surr2 carries the other half
        float g_logp;
        const bool inside = (ratio >= 1.f - g.clip) && (ratio <= 1.f + g.clip);
        if (inside) g_logp = -(ad * ratio);
        else if (surr1 < surr2) g_logp = -(ad * ratio);
        else if (surr1 == surr2) g_logp = -(ad * ratio) * 0.5f;
        else g_logp = 0.f;
        // gradient wrt the clamped probabilities: logp_a -> 1 / p_a, -ent_coef * H -> ent_coef * (log p + 1); the clamp passes it where 1e-11 <= s <= 1
        const float g_a = g_logp / pa;
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < PER; j++) {
            float gj = (sub + 4 * j < A) ? fmaf(g.ent_coef, lp[j], g.ent_coef) : 0.f;
            if (sub + 4 * j == a) gj += g_a;
            gj = (s[j] >= 1e-11f && s[j] <= 1.f) ? gj : 0.f;
            lp[j] = gj; dot = fmaf(gj, s[j], dot);
        }
        dot = quad_sum(dot);
        const float c = valid ? g.inv_temp * g.scale * (g.loss_scale_dev ? *g.loss_scale_dev : g.loss_scale) : 0.f;
        short* drow = buf1 + row * LD + sub;
#pragma unroll
        for (int j = 0; j < PER; j++) drow[4 * j] = f2s<HALF>(s[j] * (lp[j] - dot) * c);      // (s = 0 beyond the A logits: zeros)
        if (g.metrics) {
            const bool cnt = valid && sub == 0;
            const float lr = lpa - olp;
            float m_ent = cnt ? ent : 0.f, m_kl = cnt ? (ratio - 1.f) - lr : 0.f, m_clip = (cnt && fabsf(ratio - 1.f) > g.clip) ? 1.f : 0.f, m_ratio = cnt ? ratio : 0.f;
            m_ent = wave_sum_(m_ent); m_kl = wave_sum_(m_kl); m_clip = wave_sum_(m_clip); m_ratio = wave_sum_(m_ratio);
            if (lane == 0) { red[w][0] = m_ent; red[w][1] = m_kl; red[w][2] = m_clip; red[w][3] = m_ratio; }
        }
    } else {
        float sq = 0.f;
        if (tid < R) {
            const int gm = m0 + tid;
            const bool valid = gm < g.rows;
            const float d = zf[tid * LDF] - pre_tgt;
            const float grd = 2.f * d * g.scale * (g.loss_scale_dev ? *g.loss_scale_dev : g.loss_scale);
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
            v[0] = valid ? f2s<HALF>(grd) : (short)0;
            short* drow = buf1 + tid * LD;
            *reinterpret_cast<bf16x8*>(drow) = v;
#pragma unroll
            for (int c = 8; c < OUTP; c += 8) *reinterpret_cast<bf16x8*>(drow + c) = zero;
            sq = valid ? d * d : 0.f;
        }
        if (g.metrics) { sq = wave_sum_(sq); if (lane == 0) red[w][0] = sq; }
    }
    __syncthreads();
    stamp(5);
    if (g.metrics) {
        if (POLICY) { if (tid < 4) { float t = 0.f; for (int k = 0; k < NWAVE; k++) t += red[k][tid]; atomicAdd(&g.metrics[tid], t); } }
        else if (tid == 0) { float t = 0.f; for (int k = 0; k < NWAVE; k++) t += red[k][0]; atomicAdd(&g.metrics[4], t); }
    }
    // ---- backward: dZ_(i-1) = (dZ_i . W_i) where layer (i-1)'s activation was positive ----
    Fr<NKH> g2, g1;
    load_fr(g2, n.wtf[2], w, lane);
    if (!FZ_DBG(1)) copy_out<OUTP>(buf1, n.dy[3], m0, g.rows, tid);
    if (!FZ_DBG(2)) mma_tiles<NKO, HALF>(g3, buf1, lane, zero16(), acc);
    epilogue_dx<HALF>(acc, mask2, w, buf0, lane);
    __syncthreads();
    stamp(6);
    load_fr(g1, n.wtf[1], w, lane);
    if (!FZ_DBG(1)) copy_out<H>(buf0, n.dy[2], m0, g.rows, tid);
    if (!FZ_DBG(2)) mma_tiles<NKH, HALF>(g2, buf0, lane, zero16(), acc);
    epilogue_dx<HALF>(acc, mask1, w, buf1, lane);
    __syncthreads();
    stamp(7);
    if (!FZ_DBG(1)) copy_out<H>(buf1, n.dy[1], m0, g.rows, tid);
    if (!FZ_DBG(2)) mma_tiles<NKH, HALF>(g1, buf1, lane, zero16(), acc);
    epilogue_dx<HALF>(acc, mask0, w, buf0, lane);
    __syncthreads();
    stamp(8);
    if (!FZ_DBG(1)) copy_out<H>(buf0, n.dy[0], m0, g.rows, tid);
    stamp(9);
#ifdef FZ_PROF
    if (g.prof && tid == 0) for (int k = 0; k < 10; k++) atomicAdd(&g.prof[(POLICY ? 0 : 16) + k], t_st[k + 1] - t_st[k]);
#endif
}

// grid (stripes, 2): blockIdx.y = 0 policy, 1 critic
template <int K0P, int OUTP, bool HALF = false>
__global__ void __launch_bounds__(512) k_ppo_fwd_bwd(Args g) {
    extern __shared__ __attribute__((aligned(16))) short fz_smem[];
    __shared__ float red[NWAVE][4];
    short* buf0 = fz_smem;
    short* buf1 = fz_smem + R * LD;
    if (blockIdx.y == 0) stripe_body<K0P, OUTP, true, HALF>(g, g.net[0], buf0, buf1, red);
    else stripe_body<K0P, 32, false, HALF>(g, g.net[1], buf0, buf1, red);
}

// ---- the value pass (Learner::AddNewExperience, Learner.cpp:296-316: valueNet->Forward over every row of the iteration) -----------------------
// The critic's forward chain of a 128-row stripe with the activations in LDS, as in k_ppo_fwd_bwd, and nothing written but the values:
// was k_mlp_infer (32 rows per workgroup, the 350 KB of weights streamed once per 32 rows).
struct ValueArgs { const float* obs; int rows; int D; const short* wf[4]; const float* bias[4]; float* values; };

template <int K0P, bool HALF = false>
__global__ void __launch_bounds__(512) k_value_stripe(ValueArgs g) {
    extern __shared__ __attribute__((aligned(16))) short fz_smem[];
    constexpr int NK0 = K0P / 16, NKH = H / 16;
    short* buf0 = fz_smem;
    short* buf1 = fz_smem + R * LD;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.x * R;
    Fr<NK0> f0; load_fr(f0, g.wf[0], w, lane);
    float* const bias_lds = reinterpret_cast<float*>(fz_smem + 2 * R * LD);
    stage_bias(bias_lds, g.bias, 1, tid);
    {   // rows -> bf16 (the gather of stripe_body without an index list)
        struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
        const int row = tid >> 2, sub = tid & 3, gm = m0 + row;
        const bool valid = gm < g.rows;
        const float* src = g.obs + (size_t)(valid ? gm : g.rows - 1) * g.D;
        const int full = g.D >> 3;
        F4 va[K0P / 32], vb[K0P / 32]; float tail[8];
#pragma unroll
        for (int jj = 0; jj < K0P / 32; jj++) {
            const int pi = sub + 4 * jj, pc = pi < full ? pi : full - 1;
            va[jj] = *reinterpret_cast<const F4*>(src + pc * 8); vb[jj] = *reinterpret_cast<const F4*>(src + pc * 8 + 4);
        }
#pragma unroll
        for (int q = 0; q < 8; q++) tail[q] = src[full * 8 + q < g.D ? full * 8 + q : g.D - 1];
#pragma unroll
        for (int jj = 0; jj < K0P / 32; jj++) {
            const int pi = sub + 4 * jj, c0 = pi * 8;
            float x[8] = {va[jj].x, va[jj].y, va[jj].z, va[jj].w, vb[jj].x, vb[jj].y, vb[jj].z, vb[jj].w};
            bf16x8 v;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const float e = pi < full ? x[q] : ((pi == full && c0 + q < g.D) ? tail[q] : 0.f);
                v[q] = valid ? f2s<HALF>(e) : (short)0;
            }
            *reinterpret_cast<bf16x8*>(buf0 + row * LD + c0) = v;
        }
    }
    __syncthreads();
    f32x16 acc[RT];
    Fr<NKH> f1, f2, f3;
    load_fr(f1, g.wf[1], w, lane);
    f32x16 bi = load_bias(bias_lds, w, lane);
    mma_tiles<NK0, HALF>(f0, buf0, lane, bi, acc);
    epilogue_hidden<HALF>(acc, w, buf1, lane);
    __syncthreads();
    load_fr(f2, g.wf[2], w, lane);
    bi = load_bias(bias_lds + H, w, lane);
    mma_tiles<NKH, HALF>(f1, buf1, lane, bi, acc);
    epilogue_hidden<HALF>(acc, w, buf0, lane);
    __syncthreads();
    load_fr(f3, g.wf[3], 0, lane);                   // the last layer has ONE feature block (the value in column 0): row tile w of wavefronts 0..3
    bi = load_bias(bias_lds + 2 * H, w, lane);
    mma_tiles<NKH, HALF>(f2, buf0, lane, bi, acc);
    epilogue_hidden<HALF>(acc, w, buf1, lane);
    __syncthreads();
    if (w < RT) {
        f32x16 a1 = zero16();
        const short* brow = buf1 + (w * 32 + (lane & 31)) * LD + 8 * (lane >> 5);
#pragma unroll
        for (int s = 0; s < NKH; s++) a1 = mma16<HALF>(f3.f[s], *reinterpret_cast<const bf16x8*>(brow + s * 16), a1);
        // feature 0 of the block = register 0 of the lanes 0..31 (features 4 (lane >> 5) + ...: the lower half holds feature 0)
        const int gm = m0 + w * 32 + (lane & 31);
        if (lane < 32 && gm < g.rows) g.values[gm] = a1[0] + bias_lds[3 * H];
    }
}

// ---- grouped dW ---------------------------------------------------------------------------------------------------------------------------
constexpr int DBK = 32;          // rows per step
constexpr int DLD = 288;         // LDS row of the 256-wide k-major tiles (576 B = 16 banks mod 64 per k row: the transposing read's 4 rows x 4 column-quads x 2 groups hit 64 distinct banks)
constexpr size_t DW_SMEM_BYTES = (size_t)2 * 2 * DBK * DLD * sizeof(short);

struct DwLayer {
    const short* Y; int ldy;     // dL/d(pre-activation) [rows][ldy] bf16 (columns Mo..ldy zero)
    const short* X; int ldx;     // layer input [rows][ldx] bf16 (columns No..ldx zero)
    int Mo, No;                  // dW is [Mo][No]
    float* dW; float* db;
};
struct DwArgs {
    DwLayer L[2][4]; int rows; int slab; int debug; int row0;   // rows [row0, rows)   // debug & 1 (tools): no flush
    // deterministic-gradient mode (rlgpu_learner_set_deterministic): a slab's dW / db go to ITS OWN copy of the flat gradient layout, with plain
    // stores -- partial[slab][offset of the element in the gradient buffer] -- and k_dw_reduce adds the slabs up in slab order.  null: fp32 atomics
    // straight into the gradient buffer, in whatever order the slabs finish.
    float* partial; size_t partial_stride; const float* grads_base;
};

__device__ __forceinline__ bf16x8 tr_operand(const short* S, int k16, int col0, int lane) {
    // 32 (cols) x 16 (k) MFMA operand out of a k-major LDS tile: per 16-lane group a 4 (k) x 16 (col) block, delivered column-major
    const int gq = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int kb = k16 + 8 * (gq >> 1), cb = col0 + 16 * (gq & 1) + 4 * p;
    using lds_v4 = __attribute__((address_space(3))) bf16x4;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(S + (kb + q) * DLD + cb));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(S + (kb + 4 + q) * DLD + cb));
    bf16x8 r; r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3]; r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

// grid (slabs, 4 layers, 2 networks), 512 threads: wavefront w owns the dW blocks [2 (w >> 1), +2) x [4 (w & 1), +4) of 32 x 32
template <bool HALF = false>
__global__ void __launch_bounds__(512) k_dw_grouped(DwArgs g) {
    extern __shared__ __attribute__((aligned(16))) short dw_smem[];
    const DwLayer L = g.L[blockIdx.z][blockIdx.y];
    if (!L.Y) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    // slabs from the minibatch's END: the rows k_ppo_fwd_bwd wrote last are the ones the last-level cache still holds
    const int r_begin = g.row0 + (gridDim.x - 1 - blockIdx.x) * g.slab, r_end = min(g.rows, r_begin + g.slab);
    if (r_begin >= r_end) return;
    bool vi[2], vj[4];
#pragma unroll
    for (int i = 0; i < 2; i++) vi[i] = (wm * 2 + i) * 32 < L.Mo;
#pragma unroll
    for (int j = 0; j < 4; j++) vj[j] = (wn * 4 + j) * 32 < L.No;
    const bool bias_wave = wn == 0 && L.db;
    f32x16 acc[2][4], accb[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int q = 0; q < 16; q++) accb[i][q] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
    }
    constexpr short ONE = HALF ? (short)0x3C00 : (short)0x3F80;
    const bf16x8 ones = {ONE, ONE, ONE, ONE, ONE, ONE, ONE, ONE};
    // a step's operands: 32 rows x 256 columns each = 2 x 1024 pieces of 16 bytes; thread t moves pieces t and t + 512 of both.  Loads are
    // UNCONDITIONAL (row and column clamped into the matrix, the value zeroed when it is stored to LDS): with predicated loads the compiler
    // cannot count what is in flight and waits for everything before every LDS store -- no prefetch left.
    struct Regs { uint4 y[2], x[2]; };
    const int n_steps = (r_end - r_begin + DBK - 1) / DBK;
    auto load = [&](Regs& rg, int step) {
        const int st = step < n_steps ? step : n_steps - 1;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int idx = tid + j * 512, row = idx >> 5, c = (idx & 31) * 8;
            const int gr = min(r_begin + st * DBK + row, r_end - 1);
            rg.y[j] = *reinterpret_cast<const uint4*>(L.Y + (size_t)gr * L.ldy + min(c, L.ldy - 8));
            rg.x[j] = *reinterpret_cast<const uint4*>(L.X + (size_t)gr * L.ldx + min(c, L.ldx - 8));
        }
    };
    auto store = [&](const Regs& rg, int step) {
        short* Ys = dw_smem + (step & 1) * 2 * DBK * DLD;
        short* Xs = Ys + DBK * DLD;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int idx = tid + j * 512, row = idx >> 5, c = (idx & 31) * 8;
            const bool in = r_begin + step * DBK + row < r_end;
            // (component-wise: a select between two uint4 VALUES goes through a scratch array)
            const bool oy = in && c < L.ldy, ox = in && c < L.ldx;
            *reinterpret_cast<uint4*>(Ys + row * DLD + c) = make_uint4(oy ? rg.y[j].x : 0u, oy ? rg.y[j].y : 0u, oy ? rg.y[j].z : 0u, oy ? rg.y[j].w : 0u);
            *reinterpret_cast<uint4*>(Xs + row * DLD + c) = make_uint4(ox ? rg.x[j].x : 0u, ox ? rg.x[j].y : 0u, ox ? rg.x[j].z : 0u, ox ? rg.x[j].w : 0u);
        }
    };
    auto compute = [&](int stage) {
        const short* Ys = dw_smem + stage * 2 * DBK * DLD;
        const short* Xs = Ys + DBK * DLD;
#pragma unroll
        for (int ks = 0; ks < DBK; ks += 16) {
            bf16x8 a[2], b[4];
#pragma unroll
            for (int i = 0; i < 2; i++) if (vi[i]) a[i] = tr_operand(Ys, ks, (wm * 2 + i) * 32, lane);
#pragma unroll
            for (int j = 0; j < 4; j++) if (vj[j]) b[j] = tr_operand(Xs, ks, (wn * 4 + j) * 32, lane);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                if (!vi[i]) continue;
#pragma unroll
                for (int j = 0; j < 4; j++) if (vj[j]) acc[i][j] = mma16<HALF>(a[i], b[j], acc[i][j]);
                if (bias_wave) accb[i] = mma16<HALF>(a[i], ones, accb[i]);
            }
        }
    };
    Regs r0, r1, r2;
    load(r0, 0); load(r1, 1);
    store(r0, 0);
    __syncthreads();
    // step t computes LDS stage t & 1 while step t + 1 waits in registers and step t + 2 is in flight
    for (int t = 0; t < n_steps; t += 3) {
        load(r2, t + 2);
        compute(t & 1);
        store(r1, t + 1);
        __syncthreads();
        if (t + 1 >= n_steps) break;
        load(r0, t + 3);
        compute((t + 1) & 1);
        store(r2, t + 2);
        __syncthreads();
        if (t + 2 >= n_steps) break;
        load(r1, t + 4);
        compute((t + 2) & 1);
        store(r0, t + 3);
        __syncthreads();
    }
#ifdef FZ_DEBUG   /* experiments only, like FZ_DBG above: a release library has no way to leave the flush out */
    if (g.debug & 1) return;
#endif
    // C layout: m = (q & 3) + 8 (q >> 2) + 4 (lane >> 5) down the block's dW rows, lane & 31 along its columns
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (!vi[i]) continue;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!vj[j]) continue;
            const int gn = (wn * 4 + j) * 32 + (lane & 31);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int gm = (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (gm < L.Mo && gn < L.No) {
                    float* dst = &L.dW[(size_t)gm * L.No + gn];
                    if (g.partial) g.partial[(size_t)blockIdx.x * g.partial_stride + (size_t)(dst - g.grads_base)] = acc[i][j][q];
                    else atomicAdd(dst, acc[i][j][q]);
                }
            }
        }
        if (bias_wave && (lane & 31) == 0) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const int gm = (wm * 2 + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
                if (gm < L.Mo) {
                    float* dst = &L.db[gm];
                    if (g.partial) g.partial[(size_t)blockIdx.x * g.partial_stride + (size_t)(dst - g.grads_base)] = accb[i][q];
                    else atomicAdd(dst, accb[i][q]);
                }
            }
        }
    }
}

// deterministic-gradient mode: grads[i] += partial[0][i] + partial[1][i] + ... in slab order (one thread per parameter of the layers the dW launch covered)
__global__ void k_dw_reduce(const float* partial, size_t stride, int n_slabs, float* grads, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < n_slabs; k++) s += partial[(size_t)k * stride + (size_t)i];
    grads[i] += s;
}

}  // namespace fused
