// rlgpu_env.hip — the batched arena stepper on gfx950: one wavefront lane per env, SoA resident state,
// LDS-staged BVH top levels, obs/reward/done rows written straight into caller-provided device buffers.
// Implements the rlgpu_env_* half of include/rlgpu.h.  The per-env algorithm lives in arena_*.h (restating
// RocketSim's Arena::Step and RLGymSim's Gym::Step; citations there).
//
// Build: hipcc --offload-arch=gfx950 -ffp-contract=off (same contraction setting as the host port so the two
// builds agree to libm rounding).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <dirent.h>
#include <fstream>
#include <algorithm>

#ifndef RLG_WAVES_PER_BLOCK
#define RLG_WAVES_PER_BLOCK 1
#endif
#define RLG_TICKWORK_OVERLAY 1   /* arena_step.h TickWork: the car tick context shares its LDS bytes with the contact list */
#define RLG_QUEUE_LEAVES 1       /* arena_world.h CollideQueue: candidates are kept as BVH leaves, a slot's word is derived */
#define RLG_WAVES_PER_BLOCK_DEFAULTED RLG_WAVES_PER_BLOCK
#ifdef RLG_TICK_PROFILE
// profiler build only (make PROFILE=1 -> librlgpu_prof.so): per-workgroup phase accumulators fed by RLG_PROF(i) in arena_step.h
__shared__ unsigned long long g_prof[12];
__device__ unsigned long long g_step_prof[16 * 4096];
__device__ int g_dbg[64];
#define RLG_DBG_COUNT(i) atomicAdd(&g_dbg[i], 1)   // scratch for ad-hoc device introspection (profiler build only; read with rlgpu_env_debug_ints)   // k_env_step's buckets per workgroup (first 4096 workgroups)
// per-item clocks of the narrowphase (collide_run_item): g_dbg[16 + 4 t] items of type t, [17 + 4 t] of them with a contact, [18 + 4 t] cycles / 64, [19 + 4 t] the slowest
// GJK statistics (arena_gjk.h hooks): g_dbg[32] hitbox_triangle calls, [33] GJK runs, [34] finished runs, [35] sum of their iteration counts, [36] the most
// iterations of one run, [37..40] simplex updates with 1..4 vertices, [41..63] runs by iteration count (capped)
#ifdef RLG_GJK_COUNT   // (-DRLG_GJK_COUNT: the atomics sit inside the GJK loop and distort every clock; off for timing runs)
#define RLG_GJK_STATS(i, v) do { if ((i) == 2) { atomicAdd(&g_dbg[34], 1); atomicAdd(&g_dbg[35], (int)(v)); atomicMax(&g_dbg[36], (int)(v)); atomicAdd(&g_dbg[41 + ((v) < 22 ? (int)(v) : 22)], 1); } else if ((i) >= 5) atomicAdd(&g_dbg[32 + (i)], 1); else atomicAdd(&g_dbg[32 + (i)], 1); } while (0)
#endif
#define RLG_ITEM_CLOCK() __builtin_amdgcn_s_memtime()
#ifdef RLG_SPAN_CLOCKS   // (-DRLG_SPAN_CLOCKS: two contended atomics per GJK run -- at 4096 envs they triple the tick time; off unless asked for)
#define RLG_SPAN_DONE(slot, cyc) do { atomicAdd(&g_dbg[slot], (int)((cyc) >> 6)); atomicAdd(&g_dbg[(slot) + 1], 1); } while (0)   // [slot] cycles / 64, [slot + 1] spans
#else
#define RLG_SPAN_DONE(slot, cyc) ((void)0)
#endif
#define RLG_ITEM_DONE(type, n, cyc) do { atomicAdd(&g_dbg[16 + 4 * (type)], 1); if ((n) > 0) atomicAdd(&g_dbg[17 + 4 * (type)], 1); atomicAdd(&g_dbg[18 + 4 * (type)], (int)((cyc) >> 6)); atomicMax(&g_dbg[19 + 4 * (type)], (int)(cyc)); } while (0)
__shared__ unsigned long long g_prof_last;
#ifdef RLG_FINE_PROF   // (-DRLG_FINE_PROF: one bucket per phase of arena_tick_wave instead of the coarse ones; tools/fine_prof.py reads the sums from g_dbg)
__shared__ unsigned long long g_fine[64];   // [0, 32): the phases of arena_tick_wave (RLG_FPROF); [32, 64): sub-phase stamps inside the phase functions (RLG_SPROF)
__device__ unsigned int g_fine_blk[4096 * 32];   // the buckets of every workgroup of the last k_env_ticks launch, cycles / 16
#define RLG_PROF(i) ((void)0)
#define RLG_FPROF(i)                                                             \
    do {                                                                         \
        unsigned long long _t = __builtin_amdgcn_s_memtime();                    \
        if (threadIdx.x == 0) { g_fine[i] += _t - g_prof_last; g_prof_last = _t; } \
    } while (0)
#define RLG_SPROF(i) RLG_FPROF(i)
#undef RLG_ITEM_DONE
#define RLG_ITEM_DONE(type, n, cyc) ((void)0)
#undef RLG_DBG_COUNT
#define RLG_DBG_COUNT(i) ((void)0)
#else
#define RLG_PROF(i)                                                              \
    do {                                                                         \
        unsigned long long _t = __builtin_amdgcn_s_memtime();                    \
        if (threadIdx.x == 0) { g_prof[i] += _t - g_prof_last; g_prof_last = _t; } \
    } while (0)
#endif
#endif
#ifndef RLG_FPROF
#define RLG_FPROF(i) ((void)0)
#endif
#ifndef RLG_DBG_COUNT
// Product build: the narrowphase's queue overflows are counted (they are rare -- 1.8 per million env-ticks while a policy learns -- and each
// one sends its env through the inline fallback for that tick, same results): 0 BVH frontier, 1 ball region, 2 car region, 3 item queue,
// 4 result pool.  rlgpu_env_overflow_counts reads them; other counter slots (profiler build) compile to nothing.
__device__ unsigned int g_overflow[16];
#define RLG_DBG_COUNT(i) do { if ((i) < 8 || (i) == 10) atomicAdd(&g_overflow[(i) & 15], 1u); } while (0)   // (5, 6: penetration-depth queries / those that needed the full-size arena; 7: contacts lost -- never; 10: env-ticks redone with the big contact layout)
#define RLG_HAVE_OVERFLOW_COUNTS 1
#endif
// Where the penetration-depth solver (arena_epa.h: Bullet's second GJK + EPA, for hitbox-mesh contacts deeper than the collision margin)
// keeps its state on the device.  Small arenas: the frontier / query-box / candidate-slot part of every env's CollideQueue, which is dead
// from the item compaction to the next tick's candidate walk -- 14 support vertices and 34 faces each, enough for 99.8 % of the queries
// (tools/gjk_fuzz.py prints the histogram).  The lanes of a wavefront that need the solver at the same time (a ballot: only lanes that are
// in this very branch together can collide) take the wavefront's arenas in rounds of one lane per arena.  Full-size arena (Bullet's 128
// vertices / 256 faces): global memory, one per wavefront (EnvDev::epa_big), used by one lane at a time.
#define RLG_EPA_MAX_ARENAS 4
#ifdef RLG_TICK_PROFILE   /* profiler build: penetration-depth queries per workgroup of a collection launch (slot 3 of its g_step_prof row) */
#define RLG_EPA_WG_COUNT() do { if (blockIdx.x < 4096) atomicAdd(&g_step_prof[16 * blockIdx.x + 3], 1ull); } while (0)
#else
#define RLG_EPA_WG_COUNT() ((void)0)
#endif
#if defined(RLG_EXPERIMENT_EPA_LDS) && defined(__HIP_DEVICE_COMPILE__)   /* what-if build: every EPA arena is assumed to be in LDS, the full-size pass is dropped */
#define RLG_EPA_IN_LDS(ref) __builtin_assume(__builtin_amdgcn_is_shared((const void*)&(ref)))
#endif
__shared__ unsigned char* g_epa_small_ptr[RLG_WAVES_PER_BLOCK_DEFAULTED][RLG_EPA_MAX_ARENAS];
__shared__ int g_epa_small_n[RLG_WAVES_PER_BLOCK_DEFAULTED];
__shared__ unsigned char* g_epa_big_ptr[RLG_WAVES_PER_BLOCK_DEFAULTED];
__shared__ unsigned char* g_big_work_ptr;   // EnvDev::big_work / big_locks of the launch (the tick has no EnvDev at hand: one more argument is one more spill)
__shared__ uint32_t* g_big_locks_ptr;
#define RLG_EPA_LDS_V 14
#define RLG_EPA_LDS_F 34
#if defined(__HIP_DEVICE_COMPILE__)
#define RLG_EPA_ARENA_DECL \
    const int epa_wave_ = threadIdx.x >> 6; \
    EpaArena epa_small_ = epa_arena_at(g_epa_small_ptr[epa_wave_][0], RLG_EPA_LDS_V, RLG_EPA_LDS_F); \
    EpaArena epa_bigv_ = epa_arena_at(g_epa_big_ptr[epa_wave_], EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES); \
    EpaArena* epa_big_ = nullptr;   /* (the full-size arena is taken in its own serialised pass: RLG_EPA_BIG_PASS) */
// rounds over the lanes that are here together: lane of rank r takes arena r mod n in round r / n
#define RLG_EPA_SERIALIZE_BEGIN { \
    const unsigned long long pend_ = __ballot(1); \
    const int rank_ = __popcll(pend_ & ((1ull << (threadIdx.x & 63u)) - 1ull)), total_ = __popcll(pend_), n_ar_ = g_epa_small_n[epa_wave_]; \
    for (int base_ = 0; base_ < total_; base_ += n_ar_) { if (rank_ >= base_ && rank_ < base_ + n_ar_) { RLG_DBG_COUNT(5); RLG_EPA_WG_COUNT(); \
        epa_small_ = epa_arena_at(g_epa_small_ptr[epa_wave_][rank_ - base_], RLG_EPA_LDS_V, RLG_EPA_LDS_F);
#define RLG_EPA_SERIALIZE_END } } }
#ifdef RLG_EXPERIMENT_EPA_LDS
#define RLG_EPA_BIG_PASS(rc_, CALL)
#else
#define RLG_EPA_BIG_PASS(rc_, CALL) { \
    for (unsigned long long pb_ = __ballot((rc_) == EPA_ARENA_FULL && g_epa_big_ptr[epa_wave_] != nullptr); pb_; pb_ &= pb_ - 1ull) \
        if ((int)(threadIdx.x & 63u) == __ffsll((unsigned long long)pb_) - 1) { RLG_DBG_COUNT(6); epa_big_ = &epa_bigv_; CALL; } }
#endif
#define RLG_EPA_COUNT_BIG() ((void)0)
#else   // host pass of this translation unit: never executed
#define RLG_EPA_ARENA_DECL alignas(16) unsigned char epa_mem_[epa_arena_bytes(EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES)]; \
    EpaArena epa_small_ = epa_arena_at(epa_mem_, EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES); EpaArena* epa_big_ = nullptr;
#define RLG_EPA_SERIALIZE_BEGIN
#define RLG_EPA_SERIALIZE_END
#define RLG_EPA_BIG_PASS(rc_, CALL)
#define RLG_EPA_COUNT_BIG() ((void)0)
#endif
#include "../../include/rlgpu.h"
#include "arena_gym.h"
#include "rlgpu_internal.h"
#include "arena_mesh.h"

using namespace rlg;

static_assert(sizeof(RlgpuGymConfig) == sizeof(GymConfig), "C-ABI gym config must mirror rlg::GymConfig");

namespace {

constexpr int WAVE = 64;
// The stepping kernels' real calls (the tick, the inference step, the candidate walk) are to LOCAL functions: LLVM's inter-procedural register allocation
// then drops the callee-saved convention for them (no saves in the callee's prologue; the caller is told exactly what the callee clobbers and keeps what it
// needs elsewhere) -- unless a call site carries the `tail` marker, which the optimiser puts on every call that cannot see the caller's stack (the inference
// step, once its arguments stopped pointing into the kernel's frame, saved and restored 84 vector + 36 scalar registers per call: ~20 K cycles per step).
// The attribute keeps the marker off the kernels' call sites; nothing here is, or could be, a real tail call.
#define RLG_NO_TAIL_MARK __attribute__((disable_tail_calls))
// Tuning knobs (overridable with -D for experiments): LDS per workgroup decides how many workgroups (= wavefronts) share
// a CU's 160 KiB (MI355X_MICROARCH.md); RLG_WAVES_PER_SIMD is the occupancy the register allocator is asked to allow.
#ifndef RLG_LDS_BUDGET
#define RLG_LDS_BUDGET (40 * 1024)
#endif
#ifndef RLG_LDS_NODES
#define RLG_LDS_NODES 16    /* the FEWEST BVH top nodes a workgroup stages (staged_nodes<NC>() takes what its envs leave of the budget, up to 192) */
#endif
#ifndef RLG_WAVES_PER_SIMD
#define RLG_WAVES_PER_SIMD 1
#endif
#ifndef RLG_WAVES_PER_BLOCK
#define RLG_WAVES_PER_BLOCK 1
#endif
constexpr int WPB = RLG_WAVES_PER_BLOCK;   // wavefronts per workgroup: they share the staged mesh, each owns lanes_per_block / WPB envs
constexpr int LDS_BUDGET = RLG_LDS_BUDGET;
constexpr int LDS_NODES = RLG_LDS_NODES;   // BVH top levels staged per workgroup at least (the occupancy grid prunes most walks)
// Envs per wavefront at most.  1v1: BASELINE's shape is 4096 envs per GPU = ONE wavefront of four envs on each of the 1024 SIMDs; a fifth env
// per wavefront would fit the LDS but leaves a fifth of the SIMDs idle while every wavefront takes ~10 % longer (measured: 10.3 M -> 10.0 M
// agent-steps/s).  Batches of 16 K envs and more gain 13 % ticks/s with -DRLG_MAX_EPW_1V1=5.
#ifndef RLG_MAX_EPW_1V1
#define RLG_MAX_EPW_1V1 4
#endif

struct EnvDev {
    uint32_t* words;      // [n_words][n_envs]
    const BvhNode* nodes; const MeshTri* tris; int n_nodes, n_tris;
    const uint32_t* grid;
    const uint32_t* pad_tab;     // PAD_TAB_WORDS boost pad lookup words (arena_step.h:pad_table_fill)
    const float* action_table;
    GymConfig cfg;
    int n_envs;
    float* step_stats;           // rlgpu_env_enable_step_stats: {player-steps, sum |car vel| (uu/s), ball touches, airborne} accumulated by the step kernels, or null
    unsigned char* epa_big;      // [wavefronts of a step launch][EPA_BIG_BYTES]: full-size penetration-depth arenas (arena_epa.h), or null
    uint32_t* leaf_cache;        // [n_envs][NC + 1][CACHE_LEAVES]: the candidate leaves kept over the ticks of a launch (CandCache)
    RlgpuArenaState* snap_out;   // host-plugin fallback (rlgpu_env_enable_snapshots): every step's GameState source, [n_envs], or null
    unsigned char* big_work;     // [BIG_WORK_SLOTS][big_work_bytes<NC>()]: where a tick whose contacts do not fit the LDS layout is redone (tick_world_big)
    uint32_t* big_locks;         // [BIG_WORK_SLOTS] 0 = free
    float cand_fat;              // how far the candidate walk's boxes are grown (chosen by the mesh's density: see CAND_FAT above)
    // step records for plugins that run after a collection launch (rlgpu_env_enable_step_records): rec_ring [T_cap][n_envs][record words] = every step's
    // GameState source; rec_resets [n_envs * T_cap][2 + record words] = (env, step, the new episode's first state) of every episode the launch ended,
    // appended through rec_count[0]; all null when off
    uint32_t* rec_ring; uint32_t* rec_resets; unsigned int* rec_count;
};

// Everything one env touches during a step lives in LDS (state + per-tick scratch): as stack objects these
// dynamically indexed arrays (cars[], contacts, solver rows) would sit in scratch memory, and with one wavefront per
// SIMD nothing hides a ~500-cycle scratch access per array element.  Only a few lanes of a wavefront are active so
// that 4096 envs spread over all 256 CUs instead of 64 waves.
// The candidate lists of an env outlive the tick that walked the BVH for them.  The walk is done for FAT query boxes (the bodies' boxes grown
// by CAND_FAT on every side) and its leaves are kept; as long as every body's box of the tick still lies inside its fat box (and the same
// bodies take part), the kept leaves are a superset of what a fresh walk would list, in the same relative order, and the exact per-triangle
// tests that follow see to it that the tick's items -- and with them every result -- are those of a fresh walk.  Worth it because the
// walk was ~10 % of a collection launch (priced by running it twice) and most bodies move a fraction of CAND_FAT per tick.
// How fat: a fatter box is renewed less often and lists more leaves, each of which costs four candidate tests on every tick it is kept.  On the
// 180-triangle procedural arena the optimum is flat around 2.0 Bullet units (round 3: 1.0 / 2.0 / 3.0 -> 23.0 / 22.4 / 22.65 ms per launch; round
// 5: 0.5 / 1.0 / 2.0 / 4.0 -> 17.65 / 18.02 / 18.14 / 18.07 M agent-steps/s); on the 10 084-triangle tessellated arena leaves are what costs:
// 0.5 / 1.0 / 1.5 / 2.0 / 3.0 / 4.0 -> 15.44 / 15.21 / 14.86 / 14.19 / 13.02 / 12.46 M.  So the value is chosen when the mesh is set (EnvDev::cand_fat):
// RLG_CAND_FAT for meshes up to RLG_CAND_DENSE_TRIS triangles, RLG_CAND_FAT_DENSE beyond.  Results do not depend on it.
#ifndef RLG_CAND_FAT
#define RLG_CAND_FAT 2.0f          // Bullet units (100 uu)
#endif
#ifndef RLG_CAND_FAT_DENSE
#define RLG_CAND_FAT_DENSE 1.0f
#endif
#ifndef RLG_CAND_DENSE_TRIS
#define RLG_CAND_DENSE_TRIS 2000
#endif
static_assert(BALL_CAND == CAR_CAND, "one leaf capacity for every body");
// The kept leaves themselves (first triangle | count << 24, ascending = the reference's visiting order): a tick that does not walk copies
// them into its queue (CollideQueue::leaf).  Where the wavefront's envs leave LDS to spare (1v1 at four envs per wavefront) they stay in
// LDS; otherwise in global memory (EnvDev::leaf_cache, [env][body][CACHE_LEAVES]: one extra round trip per tick, which buys the third 2v2 /
// second 3v3 env of a wavefront).
template <int NC>
constexpr bool leaves_in_lds() { return NC == 2 && RLG_MAX_EPW_1V1 <= 4; }
// The box a body's candidates are collected for, and when a kept list is still good.  Any box that CONTAINS the body's query box of the tick
// (arena_world.h:body_query_box: the hitbox's box united with the four suspension rays; the ball's box) gives the same results, bit for bit --
// the exact per-triangle tests of the narrowphase and of the wheel rays filter the list.  The walk is done for the query box grown by CAND_FAT
// on every side, and the list is kept while the box cannot have left that: every point of a query box is pos + R q with a FIXED body-frame q
// (hitbox corners, ray ends), so no face of it has moved by more than
//     max_i |pos_i - pos0_i|  +  max_ij |R_ij - R0_ij| * (|q_x| + |q_y| + |q_z|)_max
// since the walk -- twelve subtractions against the remembered pose.  (Until round 5 every tick computed the exact box -- three matrix-vector
// products, four wheel transforms, the occupancy-grid loop: 7 % of an idle tick -- only to compare it with the kept one.)  Renewal at 97 % of
// CAND_FAT; the rest covers the rounding of the box arithmetic.
// -DRLG_CAND_EXACT=0 walks for the CUBE around the position that holds the query box in every orientation instead (no box arithmetic even in
// walking ticks, no rotation term).  Measured on one box, collection launch in ms, procedural arena / the 10 084-triangle arena: round 4
// 13.9 / 17.6, exact boxes 13.3 / 17.3, cubes 13.05 / 19.9 -- on a dense mesh the cube's longer lists cost more than its arithmetic saves.
// Whether a body takes part is decided as before -- `active`: its box touches an occupied grid cell -- with one more bit for the bodies whose
// box does not but whose FAT box does (`watch`): only those can become active without leaving their fat box, and only they still pay for the
// occupancy-grid test every tick.
constexpr float CAND_CUBE_CAR = 1.95f;                                       // BT, > 1.895
constexpr float CAND_CUBE_BALL = (K::BALL_RADIUS * UU2BT + 0.12f) * 1.001f;   // the ball's query box is pos +- (r + 0.08 + 0.04) (arena_world.h:ball_query_aabb)
#ifndef RLG_CAND_EXACT
#define RLG_CAND_EXACT 1   /* 1: the walk uses the body's exact query box and the renewal bound has a rotation term; 0: the cube (no box arithmetic, no rotation term, bigger lists) */
#endif
#ifndef RLG_CAND_TICKS
#define RLG_CAND_TICKS 6.0f
#endif
constexpr float CAND_TICKS = RLG_CAND_TICKS;   // ticks a body's kept list should last at its present speed (cand_growth)
constexpr float CAND_FAT_SMALL = 0.25f;   // second try of a walk whose fat boxes listed more leaves than a body keeps: a quarter of the growth
constexpr float CAND_REACH = 3.5f;   // BT; (|q_x| + |q_y| + |q_z|)_max = 3.152 (a hitbox corner: |offset| + half extents; the ray ends reach 2.39)
template <int NC>
__device__ __forceinline__ void cand_box(const Arena<NC>& A, int body, V3& lo, V3& hi) {
#if RLG_CAND_EXACT
    body_query_box(A, body, false, lo, hi);   // (only called for bodies that have one: the caller's `alive`)
#else
    const float r = body == 0 ? CAND_CUBE_BALL : CAND_CUBE_CAR;
    const V3 p = body == 0 ? A.ball.b.pos : A.cars[body - 1].b.pos;
    lo = p - v3(r, r, r); hi = p + v3(r, r, r);
#endif
}
template <int NC>
struct CandCache {
    static constexpr int NB = NC + 1;
    V3 pos0[NB]; M3 rot0[RLG_CAND_EXACT ? NC : 1];   // the poses the lists were walked for
    float fat[NB];                        // ... and how far each body's box was grown (cand_growth)
    uint32_t leaf[leaves_in_lds<NC>() ? NB : 1][leaves_in_lds<NC>() ? CACHE_LEAVES : 1];
    uint8_t n[NB];                        // leaves of body b
    uint8_t alive;                        // bit b: body b had a query box then (the ball awake, the car not demolished)
    uint8_t active;                       // bit b: ... and its box touched an occupied grid cell
    uint8_t watch;                        // bit b: ... it did not, but its fat box did
    uint8_t valid;                        // 0: walk again (cleared when a launch loads the env); 1 / 2: lists walked for boxes grown by cand_fat / by CAND_FAT_SMALL of it
};
template <int NC>
struct LaneBlock { Arena<NC> A; GymEnv<NC> G; TickWork<NC> W; CandCache<NC> C; };

// per-lane stride: an ODD number of 8-byte units, so lane l starts at bank (2 * odd * l) mod 32 -> conflict-free for <= 16 lanes
template <int NC>
constexpr size_t lane_stride() { size_t w = (sizeof(LaneBlock<NC>) + 7) / 8; return ((w % 2) ? w : w + 1) * 8; }
template <int NC>
constexpr int lanes_per_block() {
#ifdef RLG_EXPERIMENT_EPW   /* what-if builds (tools/build_variant.sh): a fixed number of envs per wavefront whatever the budgets say; only k_env_ticks is meaningful */
    if (NC == 2) return RLG_EXPERIMENT_EPW * WPB;
#endif
    // as many envs per wavefront as the workgroup's LDS budget holds, at most 8 (bank-conflict-free lane strides), one lane per wheel
    // (64 / (4 NC)), and the rows of one inference tile (a wavefront infers its own envs' agents: rlinfer::WAVE_ROWS)
    int epw = NC == 2 ? RLG_MAX_EPW_1V1 : 8;
    while (epw > 1 && ((size_t)epw * WPB * lane_stride<NC>() + (size_t)LDS_NODES * sizeof(BvhNode) + (GRID_WORDS + PAD_TAB_WORDS) * 4 > (size_t)LDS_BUDGET
                       || epw * NC * 4 > WAVE || epw * NC > rlinfer::WAVE_ROWS)) epw--;
    return epw * WPB;
}
// BVH top nodes a workgroup stages: what its envs leave of the LDS budget, in steps of 8 nodes, at most 192
template <int NC>
constexpr int staged_nodes() {
    const size_t used = (size_t)lanes_per_block<NC>() * lane_stride<NC>() + (GRID_WORDS + PAD_TAB_WORDS) * 4 + 256;   // (256: the few other __shared__ words)
    int n = used < (size_t)LDS_BUDGET ? (int)(((size_t)LDS_BUDGET - used) / sizeof(BvhNode)) / 8 * 8 : 0;
    return n > 192 ? 192 : (n < LDS_NODES ? LDS_NODES : n);
}

template <int NC>
__device__ void load_env(const uint32_t* words, int n_envs, int env, Arena<NC>& A, GymEnv<NC>& G) {
    WordReader r; r.base = words + env; r.stride = (size_t)n_envs; r.idx = 0;
    arena_visit(A, G, r);
    arena_finish_load(A);
}
template <int NC>
__device__ void store_env(uint32_t* words, int n_envs, int env, Arena<NC>& A, GymEnv<NC>& G) {
    WordWriter w; w.base = words + env; w.stride = (size_t)n_envs; w.idx = 0;
    arena_visit(A, G, w);
}

__shared__ uint32_t* g_leaf_cache;   // EnvDev::leaf_cache for the tick's candidate phase (kept out of the argument lists of the per-phase calls)
__shared__ float g_cand_fat;         // EnvDev::cand_fat
__device__ __forceinline__ MeshView stage_mesh(const EnvDev& d, BvhNode* lds_nodes, int n_stage, uint32_t* lds_grid, uint32_t* lds_pad) {
    if (threadIdx.x == 0) { g_leaf_cache = d.leaf_cache; g_cand_fat = d.cand_fat; }
    if (d.grid) for (int i = threadIdx.x; i < GRID_WORDS; i += blockDim.x) lds_grid[i] = d.grid[i];
    for (int i = threadIdx.x; i < PAD_TAB_WORDS; i += blockDim.x) lds_pad[i] = d.pad_tab[i];
    int n_fast = d.n_nodes < n_stage ? d.n_nodes : n_stage;
    // 32-byte nodes copied as 2 x 16-byte vectors per lane: coalesced global reads, conflict-free ds_write_b128
    const float4* src = reinterpret_cast<const float4*>(d.nodes);
    float4* dst = reinterpret_cast<float4*>(lds_nodes);
    for (int i = threadIdx.x; i < n_fast * 2; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    MeshView mv; mv.nodes = d.nodes; mv.tris = d.tris; mv.nodes_fast = lds_nodes; mv.n_nodes = d.n_nodes; mv.n_tris = d.n_tris; mv.n_fast = n_fast;
    mv.grid = d.grid ? lds_grid : nullptr;
    mv.bp = d.grid ? d.grid + GRID_WORDS : nullptr;
    return mv;
}

// ---- wave-cooperative tick ------------------------------------------------------------------------------
// A workgroup is ONE wavefront that owns LANES envs.  The tick's phases (arena_step.h) have different widths: one work
// item per car, per wheel, or per env.  Every lane of the wave walks through the same phase sequence and picks up the
// work item its lane id maps to, so the 8 suspension rays of a 1v1 env run on 8 lanes instead of 8 times in a row on one.
// State and scratch sit in LDS; a wave's LDS operations execute in order, so phases only need a compiler-level fence.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroups of several wavefronts (RLG_WAVES_PER_BLOCK > 1) can keep their wavefronts in step: all of a CU's wavefronts then run the same
// phase's code at the same time, which is what an instruction cache shared by the CU's wavefronts wants (one tick is ~330 KB of code).
// RLG_PHASE_BARRIER: 0 none, 1 one s_barrier per tick, 2 one per phase.  Only at points every wavefront of the workgroup passes equally often.
#ifndef RLG_PHASE_BARRIER
#define RLG_PHASE_BARRIER 0
#endif
__device__ __forceinline__ void phase_sync(int level) {
    if (WPB > 1 && RLG_PHASE_BARRIER >= level) __builtin_amdgcn_s_barrier();
}

template <int NC>
__device__ __forceinline__ LaneBlock<NC>& lane_block(unsigned char* lane_mem, int e) {
    return *reinterpret_cast<LaneBlock<NC>*>(lane_mem + (size_t)e * lane_stride<NC>());
}

// ---- wave-cooperative state load / store ---------------------------------------------------------------------
// The resident layout is words[w][env].  A lane walking the field visitor straight over global memory issues ~310 dependent
// 4-byte loads (36 us per step).  Instead all 64 lanes copy the wavefront's envs' words into LDS first -- 16 B contiguous per
// word row, ~20 independent loads per lane -- into each env's TickWork area, which is dead outside the ticks; the visitor then
// runs over LDS.  Stores mirror that.
template <int NC>
__device__ __forceinline__ uint32_t* word_stage(unsigned char* wmem, int e) {
    static_assert(sizeof(TickWork<NC>) >= arena_num_words<NC>() * 4, "TickWork doubles as the word staging area");
    return reinterpret_cast<uint32_t*>(&lane_block<NC>(wmem, e).W);
}
// (`words` / `n_envs`: EnvDev's, by value -- a reference to a kernel's argument struct that escapes into a real call puts the struct into scratch memory)
template <int NC>
__device__ void load_envs_wave(const uint32_t* words, int n_envs, unsigned char* wmem, int env0, int n_valid, int lane) {
    constexpr int NW = (int)arena_num_words<NC>(), EPW = lanes_per_block<NC>() / WPB;
    for (int idx = lane; idx < NW * EPW; idx += WAVE) {
        const int w = idx / EPW, e = idx % EPW;
        if (e < n_valid) word_stage<NC>(wmem, e)[w] = words[(size_t)w * n_envs + env0 + e];
    }
    wave_sync();
    if (lane < n_valid) {
        LaneBlock<NC>& S = lane_block<NC>(wmem, lane);
        WordReader r; r.base = word_stage<NC>(wmem, lane); r.stride = 1; r.idx = 0;
        arena_visit(S.A, S.G, r);
        arena_finish_load(S.A);
    }
    wave_sync();
}
template <int NC>
__device__ void store_envs_wave(uint32_t* words, int n_envs, unsigned char* wmem, int env0, int n_valid, int lane) {
    constexpr int NW = (int)arena_num_words<NC>(), EPW = lanes_per_block<NC>() / WPB;
    wave_sync();
    if (lane < n_valid) {
        LaneBlock<NC>& S = lane_block<NC>(wmem, lane);
        WordWriter w; w.base = word_stage<NC>(wmem, lane); w.stride = 1; w.idx = 0;
        arena_visit(S.A, S.G, w);
    }
    wave_sync();
    for (int idx = lane; idx < NW * EPW; idx += WAVE) {
        const int w = idx / EPW, e = idx % EPW;
        if (e < n_valid) words[(size_t)w * n_envs + env0 + e] = word_stage<NC>(wmem, e)[w];
    }
}

// the envs one wavefront of the workgroup owns
struct WaveSlot { int lane, env0, n_valid; unsigned char* mem; };
template <int NC>
__device__ __forceinline__ WaveSlot wave_slot(unsigned char* lane_mem, int n_envs) {
    constexpr int EPW = lanes_per_block<NC>() / WPB;
    static_assert(EPW >= 1 && EPW * WPB == lanes_per_block<NC>(), "envs per workgroup must split evenly over its wavefronts");
    static_assert(EPW * NC * 4 <= WAVE, "one lane per wheel must fit the wavefront");
    const int wave = threadIdx.x >> 6;
    WaveSlot s;
    s.lane = threadIdx.x & 63;
    s.env0 = ((int)blockIdx.x * WPB + wave) * EPW;
    int left = n_envs - s.env0;
    s.n_valid = left < 0 ? 0 : (left < EPW ? left : EPW);
    s.mem = lane_mem + (size_t)wave * EPW * lane_stride<NC>();
    return s;
}

constexpr size_t EPA_BIG_BYTES = (epa_arena_bytes(EPA_BT_MAX_VERTICES, EPA_BT_MAX_FACES) + 63) & ~(size_t)63;
// the wavefront's penetration-depth arenas (see the top of this file): called once per launch by every kernel that ticks
template <int NC>
__device__ __forceinline__ void epa_arenas_setup(const EnvDev& d, unsigned char* wmem) {
    using Q = CollideQueue<NC>; using TW = TickWork<NC>;
    // free during the narrowphase: the queue's walk part (frontier, boxes, leaves, pairs are all consumed by the item compaction), what the
    // solver rows add to the union behind it, and the solver bodies
    static_assert(offsetof(TW, B) + sizeof(((TW*)nullptr)->B) - (offsetof(TW, Q) + offsetof(Q, frontier)) >= epa_arena_bytes(RLG_EPA_LDS_V, RLG_EPA_LDS_F) + 16,
                  "the small EPA arena borrows the stretch from a CollideQueue's frontier to the end of the solver bodies");
    static_assert(offsetof(Q, frontier) % 4 == 0, "arena alignment");
    if ((threadIdx.x & 63) == 0) {
        const int wave = threadIdx.x >> 6;
        constexpr int EPW = lanes_per_block<NC>() / WPB, NA = EPW < RLG_EPA_MAX_ARENAS ? EPW : RLG_EPA_MAX_ARENAS;
        for (int k = 0; k < NA; k++) g_epa_small_ptr[wave][k] = reinterpret_cast<unsigned char*>(&lane_block<NC>(wmem, k).W.Q.frontier[0][0]);
        g_epa_small_n[wave] = NA;   // (the TickWork areas of a wavefront's empty env slots are as good as any)
        g_epa_big_ptr[wave] = d.epa_big ? d.epa_big + ((size_t)blockIdx.x * WPB + wave) * EPA_BIG_BYTES : nullptr;
        g_big_work_ptr = d.big_work; g_big_locks_ptr = d.big_locks;   // (every wavefront's lane 0 stores the same two words)
    }
}

// A tick whose contacts do not fit the env's LDS layout (arena_contact.h: a further mesh object with points on one body, a car-car point beyond the
// pair pool, more contacts than solver rows -- two such env-ticks in 393 M of learned 3v3, profiles/r04h_soak.txt) is redone with the big layout
// (arena_step.h:world_step_finish_big), whose TickWork is far too big for LDS: a small pool of them in global memory, one slot taken by the
// wavefront for as long as it needs it.  Behind the TickWork in a slot: a small EPA arena -- the LDS ones borrow bytes that hold other envs'
// solver bodies by now.
constexpr int BIG_WORK_SLOTS = 64;
template <int NC> constexpr size_t big_work_bytes() { return ((sizeof(TickWork<NC, 1>) + 63) & ~(size_t)63) + ((epa_arena_bytes(RLG_EPA_LDS_V, RLG_EPA_LDS_F) + 63) & ~(size_t)63); }
// the envs of this wavefront that need it, one after the other, each on its own env lane.  Called by all lanes.
template <int NC>
__device__ __noinline__ void tick_world_big(unsigned char* lane_mem, int n_valid, MeshView mv, TickEvents& ev) {
    const int tid = threadIdx.x & 63, wave = threadIdx.x >> 6;
    LaneBlock<NC>& Se = lane_block<NC>(lane_mem, tid < n_valid ? tid : 0);
    unsigned long long todo = __ballot(tid < n_valid && Se.W.needs_big != 0);
    int slot = 0;
    if (tid == 0) {   // (holders of a slot wait for nothing: whoever spins here gets one)
        slot = (int)((blockIdx.x * WPB + wave) % BIG_WORK_SLOTS);
        while (atomicCAS(&g_big_locks_ptr[slot], 0u, 1u) != 0u) { slot = slot + 1 == BIG_WORK_SLOTS ? 0 : slot + 1; __builtin_amdgcn_s_sleep(8); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // pairs with the previous holder's __threadfence + atomicExch: nothing of its TickWork is read stale from another CU's cache
    }
    slot = __builtin_amdgcn_readfirstlane(slot);
    unsigned char* const mem = g_big_work_ptr + (size_t)slot * big_work_bytes<NC>();
    TickWork<NC, 1>& Wb = *reinterpret_cast<TickWork<NC, 1>*>(mem);
    unsigned char* const keep_ptr = g_epa_small_ptr[wave][0]; const int keep_n = g_epa_small_n[wave];
    wave_sync();
    if (tid == 0) { g_epa_small_ptr[wave][0] = mem + ((sizeof(TickWork<NC, 1>) + 63) & ~(size_t)63); g_epa_small_n[wave] = 1; }
    wave_sync();
    for (; todo; todo &= todo - 1ull) {
        if (tid == __ffsll(todo) - 1) {
            RLG_DBG_COUNT(10);
            world_step_finish_big(Se.A, mv, ev, Se.W.ball_asleep, Se.W.bp_moved, Wb);
            Se.W.L.n = 0; Se.W.n_normal = 0; Se.W.n_rows = 0;   // (nothing left for the solver phases of the small layout; solver_finish looks at needs_big)
        }
    }
    wave_sync();
    if (tid == 0) { g_epa_small_ptr[wave][0] = keep_ptr; g_epa_small_n[wave] = keep_n; __threadfence(); atomicExch(&g_big_locks_ptr[slot], 0u); }
    wave_sync();
}

// Phase 0b on the device: this tick's narrowphase candidates (arena_world.h:collide_build_candidates is the host form and
// defines the order).  The LPE lanes that serve one env walk the BVH breadth-first for ALL bodies of the env at once: a
// frontier entry is (body, node), one lane per entry tests the node against that body's query box; children and leaf
// triangles are appended in frontier order, so per body the sequence equals a walk of its own.  Ballots are group-local
// slices of wave-wide ones, so every loop below is wave-uniform.
template <int NC>
__device__ void build_candidates_wave(unsigned char* lane_mem, int n_valid, MeshView mv, uint32_t* leaf_cache, int env0) {
    constexpr int EPW = lanes_per_block<NC>() / WPB, LPE = WAVE / EPW, NB = NC + 1;
    static_assert(NB <= LPE, "one lane per body for the query boxes");
    const int tid = threadIdx.x & 63, e = tid / LPE, li = tid % LPE;
    const bool grp = e < n_valid;
    LaneBlock<NC>& S = lane_block<NC>(lane_mem, grp ? e : 0);
    CollideQueue<NC>& Q = S.W.Q;
    CandCache<NC>& C = S.C;
    uint32_t* const gleaf = leaf_cache + (size_t)(env0 + (grp ? e : 0)) * NB * CACHE_LEAVES;
    const int gshift = e * LPE;
    const unsigned long long gmask = ~0ull >> (64 - LPE);
    const unsigned long long below = (1ull << li) - 1ull;
    const bool asleep = (len2(S.A.ball.b.vel) == 0.f && len2(S.A.ball.b.angvel) == 0.f);
    const bool too_big = mv.n_nodes > 65535;   // frontier entries carry 16-bit node ids: bigger trees use the inline walk
    const float cand_fat = g_cand_fat;
    const bool all_fast = mv.n_nodes <= mv.n_fast;
    // is body b's kept list still good?  lane b of the group looks how far the body has moved since the walk (cand_box)
    bool my_active = false, my_stale = false, my_alive = false;
    V3 lo = v3(0, 0, 0), hi = v3(0, 0, 0);
    if (grp && li < NB && !too_big) {
        const Body& bd = li == 0 ? S.A.ball.b : S.A.cars[li - 1].b;
        my_alive = li == 0 ? !asleep : !(S.A.cars[li - 1].flags & CF_IS_DEMOED);
        const bool was = (C.alive >> li) & 1u;
        if (RLG_UNLIKELY(!C.valid || was != my_alive)) my_stale = true;
        else if (my_alive) {
            const V3 p = bd.pos, p0 = C.pos0[li];
            float moved = fmaxf(fabsf(p.x - p0.x), fmaxf(fabsf(p.y - p0.y), fabsf(p.z - p0.z)));
#if RLG_CAND_EXACT
            if (li > 0) {   // every point of a car's query box is pos + R q with a fixed q: a face moves by at most |dpos| + max |dR_ij| * (|q_x| + |q_y| + |q_z|)_max
                const M3 r = bd.rot, r0 = C.rot0[li - 1];
                const float turned = fmaxf(fmaxf(fmaxf(fabsf(r.r0.x - r0.r0.x), fabsf(r.r0.y - r0.r0.y)), fmaxf(fabsf(r.r0.z - r0.r0.z), fabsf(r.r1.x - r0.r1.x))),
                                           fmaxf(fmaxf(fabsf(r.r1.y - r0.r1.y), fabsf(r.r1.z - r0.r1.z)), fmaxf(fmaxf(fabsf(r.r2.x - r0.r2.x), fabsf(r.r2.y - r0.r2.y)), fabsf(r.r2.z - r0.r2.z))));
                moved += turned * CAND_REACH;
            }
#endif
            my_stale = !(moved <= 0.97f * (C.valid == 2 ? CAND_FAT_SMALL * C.fat[li] : C.fat[li]));   // (a NaN pose renews the list every tick; valid == 2: the list was walked for the smaller boxes)
            if (RLG_UNLIKELY(!my_stale && ((C.watch >> li) & 1u))) {   // not on any list, but close enough to the mesh to get onto one inside its fat box
                cand_box(S.A, li, lo, hi);
                my_stale = mesh_maybe_near(mv, lo, hi);
            }
        }
    }
    // one body out of its box and EVERY env of the wavefront walks again: the walk is level-synchronous over the whole wavefront anyway (an env
    // that would not have had to walk costs nothing extra), and boxes that are renewed together tend to run out together
    const bool walk = grp && !too_big && __any(my_stale);
#ifdef RLG_TICK_PROFILE
    if (grp && li == 0) RLG_DBG_COUNT(13);   // (env-ticks)
#endif
    RLG_SPROF(37);
    // a walking tick needs the boxes themselves: does the body's reach the mesh, and if not, does its fat version
    bool my_watch = false; float my_fat = cand_fat;
    if (RLG_UNLIKELY(walk && li < NB && my_alive)) {
        cand_box(S.A, li, lo, hi);
        my_active = mesh_maybe_near(mv, lo, hi);
        // how far this body's box is grown: what it travels in CAND_TICKS ticks at its present speed (a face of a car's box moves by at most
        // |v| dt + |w| dt x reach per tick), within [1/2, 2] x the mesh's value -- a resting body keeps few leaves, a fast one does not renew
        // every other tick, and the bodies of a wavefront (one stale body renews them all) run out at about the same time
        const Body& bd = li == 0 ? S.A.ball.b : S.A.cars[li - 1].b;
        const float per_tick = (len(bd.vel) + (li > 0 ? len(bd.angvel) * CAND_REACH : 0.f)) * TICK_DT;
        my_fat = fminf(fmaxf(per_tick * CAND_TICKS, 0.5f * cand_fat), 2.f * cand_fat);
        if (!my_active) my_watch = mesh_maybe_near(mv, lo - v3(my_fat, my_fat, my_fat), hi + v3(my_fat, my_fat, my_fat));
    }
    bool overflow = too_big;
    // The walk: fat boxes first; should their lists not fit (a body in a corner of a dense mesh), once more with boxes grown by a quarter of that, kept
    // like the first; should those not fit either, with the exact boxes, and that result is not kept (the env walks again on the next tick).
    for (int attempt = 0; attempt < 3; attempt++) {
        const float fat = attempt == 0 ? my_fat : (attempt == 1 ? CAND_FAT_SMALL * my_fat : 0.f);
        const bool go = walk && (attempt == 0 || overflow);
        if (RLG_LIKELY(!__any(go))) break;
#ifdef RLG_TICK_PROFILE   /* profiler build only: env-ticks that walk / whose fat walk did not fit (tools/prof_collect.py) */
        if (go && li == 0) { if (attempt == 0) RLG_DBG_COUNT(11); else if (attempt == 1) RLG_DBG_COUNT(12); else RLG_DBG_COUNT(14); }
#endif
        if (go) overflow = false;
        if (go && li < NB) {
            if (my_active) { Q.box_lo[li] = lo - v3(fat, fat, fat); Q.box_hi[li] = hi + v3(fat, fat, fat); }
            if (my_alive) { C.pos0[li] = li == 0 ? S.A.ball.b.pos : S.A.cars[li - 1].b.pos; C.fat[li] = my_fat; }
#if RLG_CAND_EXACT
            if (my_alive && li > 0) C.rot0[li - 1] = S.A.cars[li - 1].b.rot;
#endif
            C.n[li] = 0;
        }
        // level 0: the roots of the active bodies, in body order
        const unsigned long long ma = (__ballot(go && my_active) >> gshift) & gmask;
        const unsigned long long mal = (__ballot(go && my_alive) >> gshift) & gmask;
        const unsigned long long mwa = (__ballot(go && my_watch) >> gshift) & gmask;
        if (go && my_active) Q.frontier[0][__popcll(ma & below)] = (uint32_t)li << 16;
        int n = go ? __popcll(ma) : 0, cur = 0;
        int cnt_b[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) cnt_b[b] = 0;
        while (__any(n > 0)) {
            wave_sync();
            int m = 0;
            for (int c0 = 0; __any(c0 < n); c0 += LPE) {
                const int j = c0 + li;
                bool inner = false, leaf = false; int cnt = 0, first = 0, body = 0;
                if (j < n) {
                    const uint32_t ent = Q.frontier[cur][j];
                    body = (int)(ent >> 16);
                    const int ni = (int)(ent & 0xffffu);
                    BvhNode nd;
                    if (all_fast) { RLG_ASSUME_LDS(*mv.nodes_fast); nd = load_node(mv.nodes_fast + ni); }   // whole tree staged: ds_read instead of a flat load
                    else nd = mesh_node(mv, ni);
                    if (aabb_overlap(nd, Q.box_lo[body], Q.box_hi[body])) { cnt = node_count(nd); first = nd.left_or_first; inner = cnt == 0; leaf = cnt > 0; }
                }
                const unsigned long long mi = (__ballot(inner) >> gshift) & gmask;
                bool ovf = false;
                if (inner) {
                    const int pos = m + 2 * __popcll(mi & below);
                    if (pos + 2 > FRONTIER_CAP) { ovf = true; if (attempt == 2) RLG_DBG_COUNT(0); }
                    else { Q.frontier[cur ^ 1][pos] = ((uint32_t)body << 16) | (uint32_t)first; Q.frontier[cur ^ 1][pos + 1] = ((uint32_t)body << 16) | (uint32_t)(first + 1); }
                }
                m += 2 * __popcll(mi);
                // leaves, in frontier order per body (a ballot per body instead of a prefix sum over the triangle counts)
#pragma unroll
                for (int b = 0; b < NB; b++) {
                    const bool mine = leaf && body == b;
                    const unsigned long long ml = (__ballot(mine) >> gshift) & gmask;
                    if (mine) {
                        const int k = cnt_b[b] + __popcll(ml & below);
                        if (k >= CACHE_LEAVES) { ovf = true; if (attempt == 2) RLG_DBG_COUNT(1 + (b > 0)); }
                        else Q.leaf[b][k] = (uint32_t)first | ((uint32_t)cnt << 24);
                    }
                    cnt_b[b] += __popcll(ml);
                }
                if ((__ballot(ovf) >> gshift) & gmask) overflow = true;
            }
            n = overflow ? 0 : m;
            cur ^= 1;
        }
        wave_sync();
        // the leaves came level by level; the narrowphase wants them in the reference's visiting order = ascending first triangle
        // (arena_mesh.cpp): every lane ranks its leaves among the body's, then all are written back in place
        if (__any(go && !overflow)) {
            constexpr int PER = (CACHE_LEAVES + LPE - 1) / LPE;
            uint32_t mine[NB][PER]; int rank[NB][PER];
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int nl = (go && !overflow) ? cnt_b[b] : 0;
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    const int k = li + r * LPE;
                    mine[b][r] = 0; rank[b][r] = -1;
                    if (k < nl) {
                        const uint32_t v = Q.leaf[b][k]; const uint32_t f = v & 0xFFFFFFu;
                        int below_me = 0;
                        for (int j = 0; j < nl; j++) below_me += ((Q.leaf[b][j] & 0xFFFFFFu) < f) ? 1 : 0;
                        mine[b][r] = v; rank[b][r] = below_me;
                    }
                }
            }
            wave_sync();
#pragma unroll
            for (int b = 0; b < NB; b++)
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    if (rank[b][r] < 0) continue;
                    Q.leaf[b][rank[b][r]] = mine[b][r];
                    if (attempt < 2) {   // kept for the ticks that do not walk (C.valid below)
                        if constexpr (leaves_in_lds<NC>()) C.leaf[b][rank[b][r]] = mine[b][r]; else gleaf[b * CACHE_LEAVES + rank[b][r]] = mine[b][r];
                    }
                }
        }
        if (go && li == 0) {
#pragma unroll
            for (int b = 0; b < NB; b++) C.n[b] = (uint8_t)(cnt_b[b] < CACHE_LEAVES ? cnt_b[b] : CACHE_LEAVES);
            C.active = (uint8_t)ma;
            C.alive = (uint8_t)mal; C.watch = (uint8_t)mwa;
            C.valid = (uint8_t)((!overflow && attempt < 2) ? 1 + attempt : 0);
        }
        wave_sync();
    }
    RLG_SPROF(38);
    // a tick that did not walk takes the kept leaves (a block of LEAF_SLOTS candidate slots per leaf in the body's region: queue_cand)
    if (grp && !walk && !overflow) {
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const int nl = ((C.active >> b) & 1u) ? (int)C.n[b] : 0;
            for (int k = li; k < nl; k += LPE) {
                if constexpr (leaves_in_lds<NC>()) Q.leaf[b][k] = C.leaf[b][k]; else Q.leaf[b][k] = gleaf[b * CACHE_LEAVES + k];
            }
        }
    }
    wave_sync();
    RLG_SPROF(39);
    if (grp && li == 0) {
        S.W.ball_asleep = asleep;
#pragma unroll
        for (int b = 0; b < NB; b++) Q.cand_count[b] = (uint16_t)((!overflow && ((C.active >> b) & 1u)) ? (int)C.n[b] * LEAF_SLOTS : 0);
        Q.n_items = 0; Q.n_pool = 0; Q.overflow = overflow ? 1 : 0; Q.n_pairs = 0;
        for (int ci = 0; ci < NC; ci++)
            for (int ib = ci + 1; ib < NC; ib++)
                if (car_collides(S.A.cars[ci]) && car_collides(S.A.cars[ib]) && cars_maybe_touch(S.A, ci, ib)) queue_pair(Q, ci, ib);
    }
    wave_sync();
}

// `ev` is meaningful on env lanes (lane e < n_valid owns env e of the wavefront)
#ifdef RLG_INLINE_T8
#define RLG_TICK_INLINE __forceinline__
#else
#define RLG_TICK_INLINE
#endif
template <int NC>
__device__ RLG_TICK_INLINE void arena_tick_wave(unsigned char* lane_mem, int n_valid, MeshView mv, const uint32_t* pad_tab, uint32_t seed, int env0, TickEvents& ev) {
    constexpr int EPW = lanes_per_block<NC>() / WPB;
    RLG_ASSUME_LDS(*lane_mem);   // (not inlined into the step / collect kernels: without this every access below is a flat_load / flat_store)
    const int tid = threadIdx.x & 63;
    const int e_car = tid / NC, c_car = tid % NC;
    const int e_whl = tid / (4 * NC), c_whl = (tid >> 2) % NC, w_whl = tid & 3;
    const bool env_lane = tid < n_valid;
    const bool car_lane = e_car < n_valid;
    const bool whl_lane = e_whl < n_valid;
    LaneBlock<NC>& Sc = lane_block<NC>(lane_mem, car_lane ? e_car : 0);
    LaneBlock<NC>& Sw = lane_block<NC>(lane_mem, whl_lane ? e_whl : 0);
    LaneBlock<NC>& Se = lane_block<NC>(lane_mem, env_lane ? tid : 0);

    phase_sync(1);
    const bool ref_due = car_lane && car_tick_begin(Sc.A, c_car, seed, (uint32_t)(env0 + e_car));
    wave_sync();
    if (RLG_UNLIKELY(__any(ref_due))) {   // parity tests only: respawn draws from the reference's engine go in the arena's car order (arena_car.h)
        if (env_lane && Se.A.ref_engine != 0u) cars_respawn_ref_engine(Se.A);
        wave_sync();
    }
    RLG_PROF(0); RLG_FPROF(0);
    build_candidates_wave<NC>(lane_mem, n_valid, mv, g_leaf_cache, env0);
#ifdef RLG_EXPERIMENT_BFS_TWICE   // what-if build only: the candidate walk is idempotent
    build_candidates_wave<NC>(lane_mem, n_valid, mv, g_leaf_cache, env0);
#endif
    RLG_PROF(1); RLG_FPROF(1); phase_sync(2);
    // suspension rays: begin (lane per wheel) | mesh pairs (lane per ray x candidate triangle of the car) | finish (lane per wheel)
    if (whl_lane) car_wheel_ray_begin(Sw.A, c_whl, w_whl, Sw.W.ctx[c_whl]);
    wave_sync();
    RLG_FPROF(2); phase_sync(2);
    {   // all (env, car, wheel, candidate) pairs of the wavefront as ONE list over the 64 lanes (a car next to a wall has ~100 of
        // them, most cars none)
        int n_of[EPW * NC], total = 0;
#pragma unroll
        for (int q = 0; q < EPW * NC; q++) {
            const int e = q / NC, ci = q % NC;
            const LaneBlock<NC>& Sq = lane_block<NC>(lane_mem, e < n_valid ? e : 0);
            n_of[q] = (e < n_valid && !Sq.W.Q.overflow) ? car_ray_pairs(Sq.A, Sq.W.Q, ci) : 0;
            total += n_of[q];
        }
        for (int g = tid; g < total; g += WAVE) {
            int q = 0, pr = g;
#pragma unroll
            for (int k = 0; k < EPW * NC - 1; k++) if (q == k && pr >= n_of[k]) { pr -= n_of[k]; q = k + 1; }
            LaneBlock<NC>& Sr = lane_block<NC>(lane_mem, q / NC);
            car_ray_pair(Sr.A, mv, Sr.W.Q, q % NC, pr, Sr.W.ctx[q % NC]);
        }
    }
    wave_sync();
    RLG_FPROF(3); phase_sync(2);
    if (whl_lane) car_wheel_ray_finish(Sw.A, c_whl, w_whl, mv, Sw.W.Q, Sw.W.ctx[c_whl]);
    wave_sync();
    RLG_FPROF(4); phase_sync(2);
    const bool ordered = car_lane && car_needs_ordered_finish(Sc.W.ctx[c_car]);
    if (__ballot(ordered) == 0ull) {
        if (car_lane) car_pre_tick_finish(Sc.A, c_car, Sc.W.ctx[c_car]);
    } else {
        for (int k = 0; k < NC; k++) {   // a wheel stands on another car somewhere in this wave: car order matters (the env's car_order: Arena.cpp:716-812)
            if (car_lane && c_car == car_at_rank(Sc.A, k)) car_pre_tick_finish(Sc.A, c_car, Sc.W.ctx[c_car]);
            wave_sync();
        }
    }
    wave_sync();
    RLG_FPROF(5); phase_sync(2);
    constexpr int LPE = WAVE / EPW;           // lanes that serve one env in the pad / candidate / item phases
    const int e_grp = tid / LPE, l_grp = tid % LPE;
    const bool grp_lane = e_grp < n_valid;
    LaneBlock<NC>& Sg = lane_block<NC>(lane_mem, grp_lane ? e_grp : 0);
    if (grp_lane) for (int p = l_grp; p < 34; p += LPE) pad_pre_tick(Sg.A.pads[p]);
    if (env_lane) tick_world_begin(Se.A, Se.W, true);
    wave_sync();
    RLG_FPROF(6); phase_sync(2);
    {   // narrowphase (arena_step.h): lane per candidate tests + compacts, lane per item runs
        const int e_item = e_grp, l_item = l_grp;
        const bool item_lane = e_item < n_valid;
        LaneBlock<NC>& Si = lane_block<NC>(lane_mem, item_lane ? e_item : 0);
        CollideQueue<NC>& Q = Si.W.Q;
        {
            const bool live = item_lane && !Q.overflow;
            int base = 0;   // items of this env so far (same value on all its lanes)
            for (int body = 0; body <= NC + 1; body++) {   // the body regions, then the pair region
                const int region = body <= NC ? CollideQueue<NC>::region(body) : CollideQueue<NC>::PAIR_BASE;
                const int n_cand = !live ? 0 : (body <= NC ? (int)Q.cand_count[body] : Q.n_pairs);
                for (int c0 = 0; __any(c0 < n_cand); c0 += LPE) {
                    const int k = region + c0 + l_item;
                    const bool pass = c0 + l_item < n_cand && collide_test_candidate(Si.A, mv, Q, k);
                    const unsigned long long m = (__ballot(pass) >> (e_item * LPE)) & (~0ull >> (64 - LPE));
                    if (pass) {
                        const int pos = base + __popcll(m & ((1ull << l_item) - 1ull));
                        if (pos < ITEM_CAP) Q.items[pos] = unpack_cand(queue_cand(Q, k)); else { Q.overflow = 1; RLG_DBG_COUNT(3); }
                    }
                    base += __popcll(m);
                }
            }
            if (item_lane && l_item == 0 && !Q.overflow) Q.n_items = base;
        }
        wave_sync();
        RLG_FPROF(7); phase_sync(2);
        RLG_PROF(1);   // (the candidate tests count as "candidates", like the walk that listed them)
        {   // items of ALL envs of the wavefront as one list over the 64 lanes: a contact-heavy env borrows its neighbours' lanes
            int n_of[EPW], total = 0;
#pragma unroll
            for (int e = 0; e < EPW; e++) {
                const CollideQueue<NC>& Qe = lane_block<NC>(lane_mem, e < n_valid ? e : 0).W.Q;
                n_of[e] = (e < n_valid && !Qe.overflow) ? Qe.n_items : 0;
                total += n_of[e];
            }
            for (int g = tid; g < total; g += WAVE) {
                int e = 0, slot = g;
#pragma unroll
                for (int q = 0; q < EPW - 1; q++) if (e == q && slot >= n_of[q]) { slot -= n_of[q]; e = q + 1; }
                LaneBlock<NC>& Sx = lane_block<NC>(lane_mem, e);
                collide_run_item(Sx.A, mv, slot, Sx.W.Q);
            }
        }
        wave_sync();
    }
    RLG_PROF(2); RLG_FPROF(8); phase_sync(2);
    // rest of the world step: contacts (lane per body) | merge + row plan (env) | solver rows (lane per contact) | iterations (env) | integration (lane per body)
    {
        constexpr int NB = NC + 1;
        const int e_b = tid / NB, b_b = tid % NB;
        if (e_b < n_valid) { LaneBlock<NC>& Sb = lane_block<NC>(lane_mem, e_b); solver_body_contacts(Sb.A, mv, Sb.W, b_b, true); }
    }
    wave_sync();
    RLG_FPROF(9); phase_sync(2);
    {   // (a phase of its own: during the contacts an overflowing env may still run penetration-depth queries in the arenas that borrow these bytes)
        constexpr int NB = NC + 1;
        const int e_b = tid / NB, b_b = tid % NB;
        if (e_b < n_valid) { LaneBlock<NC>& Sb = lane_block<NC>(lane_mem, e_b); solver_body_setup(Sb.A, Sb.W, b_b); }
    }
    wave_sync();
    if (env_lane) solver_prepare(Se.A, mv, ev, Se.W, true);
    wave_sync();
    if (__any(env_lane && Se.W.needs_big != 0)) tick_world_big<NC>(lane_mem, n_valid, mv, ev);   // (contacts beyond the LDS layout: nothing is dropped)
    RLG_FPROF(10); phase_sync(2);
    if (grp_lane) for (int k = l_grp, n = Sg.W.L.n; k < n; k += LPE) solver_rows(Sg.A.mut, Sg.W, k);
    wave_sync();
    RLG_FPROF(11); phase_sync(2);
    if (env_lane) solver_iterate(Se.W);
    wave_sync();
    RLG_FPROF(12); phase_sync(2);
    {
        constexpr int NB = NC + 1;
        const int e_b = tid / NB, b_b = tid % NB;
        if (e_b < n_valid) { LaneBlock<NC>& Sb = lane_block<NC>(lane_mem, e_b); solver_finish(Sb.A, Sb.W, b_b); }
    }
    RLG_PROF(5);
    wave_sync();
    RLG_FPROF(13); phase_sync(2);
    if (car_lane) { tick_car_post(Sc.A, c_car); Sc.W.pad_mask[c_car] = pads_check_car(Sc.A, pad_tab, c_car); }
    wave_sync();
    RLG_FPROF(14); phase_sync(2);
    if (env_lane) for (int k = 0; k < NC; k++) { const int i = car_at_rank(Se.A, k); const uint64_t pm = Se.W.pad_mask[i]; if (pm) pads_lock(Se.A, i, pm); }
    wave_sync();
    RLG_FPROF(15); phase_sync(2);
    {   // pads that hand out boost are rare: they go through the env lane in pad order, all the others finish in parallel
        bool gives = false;
        if (grp_lane) for (int p = l_grp; p < 34; p += LPE) { if (pad_gives_boost(Sg.A.pads[p])) gives = true; else pad_post_tick(Sg.A, p); }
        if (__any(gives)) {
            wave_sync();
            if (env_lane) for (int p = 0; p < 34; p++) if (pad_gives_boost(Se.A.pads[p])) pad_post_tick(Se.A, p);
        }
    }
    wave_sync();
    RLG_FPROF(16); phase_sync(2);
    if (env_lane) tick_finish(Se.A, pad_tab, true);
    wave_sync();
    RLG_FPROF(17);
}

// per-step player statistics of the step's GameState (what the example program's step callback averages: examplemain.cpp:23-36), kept
// in registers by the env lanes and added to the env batch's totals once per launch
struct StepStats { float speed = 0.f, touches = 0.f, airborne = 0.f, count = 0.f; };
template <int NC>
__device__ void step_stats_add(StepStats& st, const Snapshot<NC>& S) {
    for (int k = 0; k < NC; k++) { st.speed += len(S.car_vel[k]); st.touches += S.touched[k] ? 1.f : 0.f; st.airborne += S.on_ground[k] ? 0.f : 1.f; st.count += 1.f; }
}
__device__ void step_stats_flush(float* out, StepStats st, int lane) {
    float v[4] = {st.count, st.speed, st.touches, st.airborne};
    for (int i = 0; i < 4; i++) {
        float x = v[i];
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        if (lane == 0 && x != 0.f) atomicAdd(&out[i], x);
    }
}

template <int NC>
__global__ RLG_NO_TAIL_MARK void __launch_bounds__(WAVE * WPB, RLG_WAVES_PER_SIMD) k_env_step(EnvDev d, const int32_t* actions, float* next_obs, float* reward, int32_t* done) {
    constexpr int LANES = lanes_per_block<NC>();
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    __shared__ BvhNode lds_nodes[staged_nodes<NC>()];
    __shared__ uint32_t lds_grid[GRID_WORDS];
    __shared__ uint32_t lds_pad[PAD_TAB_WORDS];
    MeshView mv = stage_mesh(d, lds_nodes, staged_nodes<NC>(), lds_grid, lds_pad);   // every thread of the workgroup helps staging
    const WaveSlot ws = wave_slot<NC>(lane_mem, d.n_envs);
    unsigned char* const wmem = ws.mem; const int env0 = ws.env0, n_valid = ws.n_valid;
    const bool env_lane = ws.lane < n_valid;
    const int env = env0 + (env_lane ? ws.lane : 0);
    LaneBlock<NC>& S = lane_block<NC>(wmem, env_lane ? ws.lane : 0);
    const uint32_t seed = tick_seed(d.cfg);
    const int D = obs_size<NC>(d.cfg);
    const int P = players_per_env<NC>(d.cfg);   // agent rows per env: NC, or NC / 2 in a one-team env
    // the GameState of the step (taken after tick 1) lives in the env's TickWork area, which is dead between ticks
    Snapshot<NC>& snap = *reinterpret_cast<Snapshot<NC>*>(&S.W);
    static_assert(sizeof(Snapshot<NC>) <= sizeof(TickWork<NC>), "the step's snapshot borrows the TickWork area");
    float rew[NC]; bool dn = false;
#ifdef RLG_TICK_PROFILE
    if (threadIdx.x == 0) { for (int i = 0; i < 12; i++) g_prof[i] = 0; g_prof_last = __builtin_amdgcn_s_memtime(); }
#endif
#ifdef RLG_POISON_LDS   /* test build: every env byte of the wavefront starts as 0xFF (NaN floats, -1 ints): results must not depend on what LDS held before */
    for (int i = (threadIdx.x & 63); i < (int)((lanes_per_block<NC>() / WPB) * lane_stride<NC>() / 4); i += 64) reinterpret_cast<uint32_t*>(wmem)[i] = 0xFFFFFFFFu;
    wave_sync();
#endif
    epa_arenas_setup<NC>(d, wmem);
    load_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, ws.lane);
    if (env_lane) { S.C.valid = 0; S.C.active = 0; S.C.alive = 0; S.C.watch = 0; }   // candidate lists are per launch
    if (env_lane) {
        int32_t acts[NC];
        for (int k = 0; k < P; k++) acts[k] = actions[(size_t)env * P + k];   // agent rows of this env (gym_step_begin maps them to slots)
        gym_step_begin<NC>(S.A, S.G, d.cfg, d.action_table, acts);
    }
    wave_sync();
    RLG_PROF(8);
    TickEvents ev; ev.bump_mask = 0;
    arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, seed, env0, ev);   // arena->Step(tickSkip - actionDelay) = 1 tick
    if (env_lane) dn = gym_step_after_first_tick<NC>(S.A, S.G, d.cfg, ev, d.action_table, (uint32_t)env, rew, next_obs + (size_t)env * P * D, (size_t)D, snap);
    // host plugins see what the reference's see: the arena as it stands where Gym::Step builds its GameState (Gym.cpp:81-93)
    if (d.snap_out && env_lane) arena_to_host(S.A, S.G, d.snap_out[env]);
    if (d.step_stats) { StepStats st; if (env_lane) step_stats_add<NC>(st, snap); step_stats_flush(d.step_stats, st, ws.lane); }
    wave_sync();
    RLG_PROF(9);
    for (int t = 1; t < d.cfg.tick_skip; t++) { TickEvents ev2; ev2.bump_mask = 0; arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, seed, env0, ev2); }
    RLG_PROF(6);
    if (env_lane) gym_step_end<NC>(S.A, S.G, d.cfg, (uint32_t)env, next_obs + (size_t)env * P * D, (size_t)D, dn, snap);
    RLG_PROF(10);
    if (env_lane) for (int k = 0; k < P; k++) { reward[(size_t)env * P + k] = rew[k]; done[(size_t)env * P + k] = dn ? 1 : 0; }
    store_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, ws.lane);
#ifdef RLG_TICK_PROFILE
    RLG_PROF(11);
    if (threadIdx.x == 0 && blockIdx.x < 4096) for (int i = 0; i < 12; i++) g_step_prof[16 * blockIdx.x + i] = g_prof[i];
#endif
}

// ---- fused collection: T x (policy inference + gym step) in ONE launch --------------------------------------------------------
// k_env_step ends when its slowest workgroup ends, and that is a workgroup holding an env in a contact-heavy phase: the mean workgroup
// needs about half as long (profiles/r01f_step_phase_cycles.txt), every workgroup is resident from the start, and nothing can fill
// the tail.  Stepping a whole collection phase inside one kernel removes the per-step rendezvous: a wavefront infers the actions of
// ITS OWN envs' agents (wave_infer: the policy MLP on <= 8 rows of one MFMA tile, weights streamed from L2, activations in the
// TickWork area that is dead between ticks), steps its envs, writes the experience rows of step t, and goes on to t + 1 -- slow
// phases of different envs now add up per wavefront over T steps instead of every step paying the worst of all envs.  The arena
// state also stays in LDS for the whole phase (one load / store per launch instead of per step).  Results are those of T alternations
// of rlgpu_policy_act and rlgpu_env_step: same inference arithmetic, same sampler counters, same stepper (log-probs within one ulp:
// this translation unit forbids fp contraction and the compiler's logf / expf expansions honour that).
struct CollectArgs {
    int T; int n_agents;
    float* obs;          // [T + 1][n_agents][D]; row block 0 holds the current observations
    int32_t* acts; float* logp; float* rew; int32_t* done;   // [T][n_agents]
    const rlinfer::InferPack* pack;   // the policy as the in-kernel inference reads it + the head's arguments (device memory: rlgpu_env::d_infer_pack)
    // free-running collection (rlgpu_collect_free): every wavefront goes on stepping its envs until the launch as a whole has gathered
    // `free_target` agent-steps (ThreadAgentManager::CollectTimesteps, ThreadAgentManager.cpp:16-32: the agents run free and the manager
    // takes what they have once the total is reached), at most T steps each (ThreadAgent.cpp:57-59, maxCollect); null counter = lockstep
    unsigned int* counter; unsigned int free_target; int32_t* steps_out;   // steps_out [n_envs]: gym steps env e made in this launch
    // step queue (k_env_collect_q): the ticket counter, every wavefront-group's finished steps, the number of groups
    unsigned int* q_ticket; int32_t* q_done; int q_groups;
};

// the first state of the episode a step just started (the env was reset inside the step: gym_step_end), appended for the host plugins' Reset hooks
template <int NC>
__device__ __noinline__ __attribute__((cold)) void record_reset(uint32_t* rec_resets, unsigned int* rec_count, const Arena<NC>& A, const GymEnv<NC>& G, int env, int t) {
    const unsigned int i = atomicAdd(rec_count, 1u);   // (the list has room for every env ending an episode in every step)
    uint32_t* o = rec_resets + (size_t)i * (2 + RLGPU_STEP_RECORD_WORDS(NC));
    o[0] = (uint32_t)env; o[1] = (uint32_t)t;
    write_step_record<NC>(A, G, nullptr, false, o + 2);
}

// The inference of one collection step for the wavefront's agents: actions and log-probs of step t go to the experience rows, the picked actions
// to `act_lds` (R ints in the last 64 bytes of env 0's TickWork area, read by the env lanes).  A real call with an allocation of its own: nothing of
// the tick is live across it (the state is in LDS), and inlined the MLP's ~220 registers (two weight buffers, a layer's A operands) sit on top of the
// step loop's -- the collection kernels then spill SGPRs into VGPR lanes and those VGPRs to scratch.  -DRLG_INFER_INLINE restores the inlined form.
#ifdef RLG_INFER_INLINE
#define RLG_INFER_STEP_ATTR __forceinline__
#else
#define RLG_INFER_STEP_ATTR __noinline__
#endif
template <int NC>
__device__ RLG_INFER_STEP_ATTR void infer_step_wave(const rlinfer::InferPack* pack_, const float* obs_, int32_t* acts_, float* logp_, int n_agents_, int t_, unsigned char* wmem_, int env0_, int n_valid_, int D_
#ifdef RLG_TICK_PROFILE
                                                    , unsigned long long* prof_st
#endif
                                                    ) {
    constexpr int LANES = lanes_per_block<NC>();
#ifdef RLG_EXPERIMENT_EPW
    constexpr int EPW = LANES / WPB, R = EPW * NC < rlinfer::WAVE_ROWS ? EPW * NC : rlinfer::WAVE_ROWS;
#else
    constexpr int EPW = LANES / WPB, R = EPW * NC;
#endif
    static_assert(R <= rlinfer::WAVE_ROWS, "a wavefront infers its own envs' agents in one MFMA tile");
    unsigned char* const wmem = rlinfer::uniform_ptr(wmem_);
    // the arguments of a real call arrive in vector registers; every one of these is the same on all lanes: back into scalar registers, so that the layer
    // table is read with scalar loads and the loops over layers and column blocks are scalar branches
    const rlinfer::InferPack RLINFER_CONST* const pk = (const rlinfer::InferPack RLINFER_CONST*)rlinfer::uniform_ptr(pack_);
    const float* const obs_base = rlinfer::uniform_ptr(obs_); int32_t* const acts_base = rlinfer::uniform_ptr(acts_); float* const logp_base = rlinfer::uniform_ptr(logp_);
    const int n_agents = __builtin_amdgcn_readfirstlane(n_agents_), t = __builtin_amdgcn_readfirstlane(t_), env0 = __builtin_amdgcn_readfirstlane(env0_),
              n_valid = __builtin_amdgcn_readfirstlane(n_valid_), D = __builtin_amdgcn_readfirstlane(D_);
    const auto& net = pk->net;
    RLG_ASSUME_LDS(*wmem);
    const int lane = threadIdx.x & 63;
    const size_t N = (size_t)n_agents;
    // inference scratch inside the TickWork areas (dead between ticks): two activation buffers and the picked actions
    const int buf_bytes = rlinfer::wave_buf_bytes(R, net.ld);
    unsigned char* const w0 = reinterpret_cast<unsigned char*>(&lane_block<NC>(wmem, 0).W);
    short* const buf0 = reinterpret_cast<short*>(w0);
    short* const buf1 = (EPW >= 2) ? reinterpret_cast<short*>(&lane_block<NC>(wmem, 1).W) : reinterpret_cast<short*>(w0 + buf_bytes);
    int* const act_lds = reinterpret_cast<int*>(w0 + sizeof(TickWork<NC>) - 64);   // the last 64 bytes of env 0's area (the inference buffers end before them)
    const int row0 = env0 * NC, n_rows = n_valid * NC;
    rlinfer::HeadArgs h = rlinfer::head_from_const(pk->head);
    h.call_ctr += (uint32_t)t; h.actions = acts_base + (size_t)t * N; h.logp = logp_base + (size_t)t * N;
    const float* const obs = obs_base + ((size_t)t * N + row0) * D;
    int picked[R];
#ifdef RLG_TICK_PROFILE
    rlinfer::wave_infer<R>(net, h, obs, row0, n_rows, buf0, buf1, lane, picked, prof_st);
#else
    if (net.fp32) {
        // exact-parity mode: fp32 activations; a buffer = NP parts of ceil(R / NP) rows, lent by the TickWork areas (dead between ticks):
        // EPW >= 4: in = envs 0, 1, out = envs 2, 3; EPW = 3: six thirds, two per area; EPW = 2: in = env 0's two halves, out = env 1's;
        // EPW = 1: four quarters of the one area
        constexpr int NP = EPW == 3 ? 3 : 2;
        constexpr int PB = (int)((sizeof(TickWork<NC>) - 64) / (EPW >= 4 ? 1 : (EPW >= 2 ? 2 : 4))) & ~15;
        auto area = [&](int k) { return reinterpret_cast<unsigned char*>(&lane_block<NC>(wmem, k < EPW ? k : 0).W); };
        rlinfer::F32Buf fin{}, fout{};
        auto f = [](unsigned char* p) { return reinterpret_cast<float*>(p); };
        if (EPW >= 4) { fin = {{f(area(0)), f(area(1)), nullptr}}; fout = {{f(area(2)), f(area(3)), nullptr}}; }
        else if (EPW == 3) { fin = {{f(area(0)), f(area(0) + PB), f(area(1))}}; fout = {{f(area(1) + PB), f(area(2)), f(area(2) + PB)}}; }
        else if (EPW == 2) { fin = {{f(area(0)), f(area(0) + PB), nullptr}}; fout = {{f(area(1)), f(area(1) + PB), nullptr}}; }
        else { fin = {{f(area(0)), f(area(0) + PB), nullptr}}; fout = {{f(area(0) + 2 * PB), f(area(0) + 3 * PB), nullptr}}; }
        rlinfer::wave_infer_f32<R, NP>(net, h, obs, row0, n_rows, fin, fout, lane, picked);
    } else
        rlinfer::wave_infer<R>(net, h, obs, row0, n_rows, buf0, buf1, lane, picked);
#endif
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < R; r++) act_lds[r] = picked[r];
    }
    wave_sync();
}

template <int NC>
__global__ RLG_NO_TAIL_MARK void __launch_bounds__(WAVE * WPB, RLG_WAVES_PER_SIMD) k_env_collect(EnvDev d, CollectArgs c) {
    constexpr int LANES = lanes_per_block<NC>();
#ifdef RLG_EXPERIMENT_EPW
    constexpr int EPW = LANES / WPB, R = EPW * NC < rlinfer::WAVE_ROWS ? EPW * NC : rlinfer::WAVE_ROWS;
#else
    constexpr int EPW = LANES / WPB, R = EPW * NC;
#endif
    static_assert(R <= rlinfer::WAVE_ROWS, "a wavefront infers its own envs' agents in one MFMA tile");
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    __shared__ BvhNode lds_nodes[staged_nodes<NC>()];
    __shared__ uint32_t lds_grid[GRID_WORDS];
    __shared__ uint32_t lds_pad[PAD_TAB_WORDS];
    MeshView mv = stage_mesh(d, lds_nodes, staged_nodes<NC>(), lds_grid, lds_pad);
    const WaveSlot ws = wave_slot<NC>(lane_mem, d.n_envs);
    unsigned char* const wmem = ws.mem; const int env0 = ws.env0, n_valid = ws.n_valid;
    if (n_valid == 0) return;
    const bool env_lane = ws.lane < n_valid;
    const int env = env0 + (env_lane ? ws.lane : 0);
    LaneBlock<NC>& S = lane_block<NC>(wmem, env_lane ? ws.lane : 0);
    const uint32_t seed = tick_seed(d.cfg);
    const int D = obs_size<NC>(d.cfg);
    const size_t N = (size_t)c.n_agents;
    int* const act_lds = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(&lane_block<NC>(wmem, 0).W) + sizeof(TickWork<NC>) - 64);   // the picked actions (infer_step_wave)
    const int row0 = env0 * NC, n_rows = n_valid * NC;
    Snapshot<NC>& snap = *reinterpret_cast<Snapshot<NC>*>(&S.W);   // the step's GameState, in the env's own TickWork area (dead between ticks)
#ifdef RLG_TICK_PROFILE
    const unsigned long long prof_t0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_step_prof[16 * blockIdx.x + 3] = 0;
    unsigned long long prof_infer = 0, prof_mlp = 0, prof_ticks = 0, prof_gym = 0, prof_stage = 0, prof_layer[4] = {0, 0, 0, 0};
#define RLG_CPROF_T0() const unsigned long long prof_c0_ = __builtin_amdgcn_s_memtime()
#define RLG_CPROF_ADD(acc) acc += __builtin_amdgcn_s_memtime() - prof_c0_
#else
#define RLG_CPROF_T0() ((void)0)
#define RLG_CPROF_ADD(acc) ((void)0)
#endif
#ifdef RLG_POISON_LDS   /* test build: every env byte of the wavefront starts as 0xFF (NaN floats, -1 ints): results must not depend on what LDS held before */
    for (int i = (threadIdx.x & 63); i < (int)((lanes_per_block<NC>() / WPB) * lane_stride<NC>() / 4); i += 64) reinterpret_cast<uint32_t*>(wmem)[i] = 0xFFFFFFFFu;
    wave_sync();
#endif
    epa_arenas_setup<NC>(d, wmem);
    load_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, ws.lane);
    if (env_lane) { S.C.valid = 0; S.C.active = 0; S.C.alive = 0; S.C.watch = 0; }   // candidate lists are per launch
    wave_sync();
    StepStats stats;
    int t = 0;
    for (; t < c.T; t++) {
        if (c.counter) {   // free-running: stop as soon as the launch has its agent-steps together (the value is the same on every lane: one address)
            const unsigned int have = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            if (have >= c.free_target) break;
        }
#ifdef RLG_TICK_PROFILE
        const unsigned long long prof_a = __builtin_amdgcn_s_memtime();
#endif
        // the observation rows of step t were written by this wavefront at the end of step t - 1 (or by the host before the launch)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#ifdef RLG_TICK_PROFILE
        unsigned long long prof_st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        infer_step_wave<NC>(c.pack, c.obs, c.acts, c.logp, c.n_agents, t, wmem, env0, n_valid, D, prof_st);
        prof_mlp += prof_st[0] - prof_a;
        prof_stage += prof_st[1] - prof_a;
        for (int q = 0; q < 4; q++) prof_layer[q] += prof_st[2 + q] - prof_st[1 + q];
#else
        infer_step_wave<NC>(c.pack, c.obs, c.acts, c.logp, c.n_agents, t, wmem, env0, n_valid, D);
#endif
#ifdef RLG_TICK_PROFILE
        prof_infer += __builtin_amdgcn_s_memtime() - prof_a;
#endif
        float rew[NC]; bool dn = false;
        if (env_lane) {
            int32_t acts[NC];
            for (int k = 0; k < NC; k++) acts[k] = act_lds[ws.lane * NC + k];
            gym_step_begin<NC>(S.A, S.G, d.cfg, d.action_table, acts);
        }
        wave_sync();
        TickEvents ev; ev.bump_mask = 0;
        { RLG_CPROF_T0(); arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, seed, env0, ev); RLG_CPROF_ADD(prof_ticks); }
        float* const obs_next = c.obs + ((size_t)(t + 1) * N + (size_t)env * NC) * D;
        { RLG_CPROF_T0();
        if (env_lane) dn = gym_step_after_first_tick<NC>(S.A, S.G, d.cfg, ev, d.action_table, (uint32_t)env, rew, obs_next, (size_t)D, snap);
        if (d.step_stats && env_lane) step_stats_add<NC>(stats, snap);
        if (d.rec_ring && env_lane) write_step_record<NC>(S.A, S.G, snap.touched, dn, d.rec_ring + ((size_t)t * d.n_envs + env) * RLGPU_STEP_RECORD_WORDS(NC));
        wave_sync(); RLG_CPROF_ADD(prof_gym); }
        { RLG_CPROF_T0(); for (int k = 1; k < d.cfg.tick_skip; k++) { TickEvents ev2; ev2.bump_mask = 0; arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, seed, env0, ev2); } RLG_CPROF_ADD(prof_ticks); }
        { RLG_CPROF_T0();
        if (env_lane) {
            gym_step_end<NC>(S.A, S.G, d.cfg, (uint32_t)env, obs_next, (size_t)D, dn, snap);
            for (int k = 0; k < NC; k++) { c.rew[(size_t)t * N + (size_t)env * NC + k] = rew[k]; c.done[(size_t)t * N + (size_t)env * NC + k] = dn ? 1 : 0; }
            if (RLG_UNLIKELY(dn && d.rec_resets != nullptr)) record_reset<NC>(d.rec_resets, d.rec_count, S.A, S.G, env, t);
        }
        RLG_CPROF_ADD(prof_gym); }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (c.counter && ws.lane == 0) __hip_atomic_fetch_add(c.counter, (unsigned int)n_rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_sync();
    }
    if (c.steps_out && env_lane) c.steps_out[env] = t;
    if (d.step_stats) step_stats_flush(d.step_stats, stats, ws.lane);
    store_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, ws.lane);
#ifdef RLG_TICK_PROFILE
    // profiler build: this workgroup's total and inference cycles (read back with rlgpu_env_debug_step_prof)
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_step_prof[16 * blockIdx.x + 0] = __builtin_amdgcn_s_memtime() - prof_t0; g_step_prof[16 * blockIdx.x + 1] = prof_infer; g_step_prof[16 * blockIdx.x + 2] = prof_mlp;
        g_step_prof[16 * blockIdx.x + 4] = prof_ticks; g_step_prof[16 * blockIdx.x + 5] = prof_gym; g_step_prof[16 * blockIdx.x + 6] = (unsigned long long)t;
        g_step_prof[16 * blockIdx.x + 7] = prof_stage; for (int q = 0; q < 4; q++) g_step_prof[16 * blockIdx.x + 8 + q] = prof_layer[q]; }
#endif
}

// ---- lockstep collection of a batch with MORE wavefront-groups than the device keeps resident: a queue of (step, group) tickets -------------------
// k_env_collect on such a batch (BASELINE configs[3] / [4]: 2 731 / 8 192 groups for 1 024 SIMDs) is balanced by the dispatcher in units of a whole
// group's T steps, and a group in a contact-heavy phase stays slow for all of them: the launch is 67 - 81 % of sum / slots (tools/prof_teams.py).
// Here as many wavefronts as fit the device stay for the whole launch and take tickets: ticket k = step k / groups of group k % groups.  A ticket's
// group state comes from HBM and goes back (13 KB per step of a 2v2 group: nothing), the observation rows of its previous step were written by
// whoever held ticket k - groups, which was handed out at least groups - wavefronts tickets earlier -- q_done[group] says when it is finished.
// Every wavefront that holds a ticket is running and waits for an EARLIER ticket only: no cycle.  Results are k_env_collect's, bit for bit: an env's
// steps do not depend on which wavefront runs them (the kept candidate lists are per ticket; they are supersets by construction).
template <int NC>
__global__ RLG_NO_TAIL_MARK void __launch_bounds__(WAVE * WPB, RLG_WAVES_PER_SIMD) k_env_collect_q(EnvDev d, CollectArgs c) {
    constexpr int LANES = lanes_per_block<NC>();
    constexpr int EPW = LANES / WPB, R = EPW * NC;
    static_assert(R <= rlinfer::WAVE_ROWS, "a wavefront infers its own envs' agents in one MFMA tile");
    static_assert(WPB == 1, "the step queue counts its groups as workgroups (collect_impl: q_groups = grid.x): with several wavefronts per workgroup the trailing groups would hold no env");
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    __shared__ BvhNode lds_nodes[staged_nodes<NC>()];
    __shared__ uint32_t lds_grid[GRID_WORDS];
    __shared__ uint32_t lds_pad[PAD_TAB_WORDS];
    MeshView mv = stage_mesh(d, lds_nodes, staged_nodes<NC>(), lds_grid, lds_pad);
    const int lane = threadIdx.x & 63;
    unsigned char* const wmem = lane_mem + (size_t)(threadIdx.x >> 6) * EPW * lane_stride<NC>();
    const uint32_t seed = tick_seed(d.cfg);
    const int D = obs_size<NC>(d.cfg);
    const size_t N = (size_t)c.n_agents;
    int* const act_lds = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(&lane_block<NC>(wmem, 0).W) + sizeof(TickWork<NC>) - 64);   // the picked actions (infer_step_wave)
    epa_arenas_setup<NC>(d, wmem);
    StepStats stats;
    const unsigned int total = (unsigned int)c.q_groups * (unsigned int)c.T;
    for (;;) {
        unsigned int tk = 0;
        if (lane == 0) tk = __hip_atomic_fetch_add(c.q_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tk = (unsigned int)__builtin_amdgcn_readfirstlane((int)tk);
        if (tk >= total) break;
        const int g = (int)(tk % (unsigned int)c.q_groups), t = (int)(tk / (unsigned int)c.q_groups);
        if (t > 0) while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&c.q_done[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < t) __builtin_amdgcn_s_sleep(32);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the group's state words and its observation rows of step t, written by another wavefront
        const int env0 = g * EPW;
        const int left = d.n_envs - env0, n_valid = left < EPW ? left : EPW;
        const bool env_lane = lane < n_valid;
        const int env = env0 + (env_lane ? lane : 0);
        LaneBlock<NC>& S = lane_block<NC>(wmem, env_lane ? lane : 0);
        const int row0 = env0 * NC, n_rows = n_valid * NC;
        Snapshot<NC>& snap = *reinterpret_cast<Snapshot<NC>*>(&S.W);
        load_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, lane);
        if (env_lane) { S.C.valid = 0; S.C.active = 0; S.C.alive = 0; S.C.watch = 0; }   // candidate lists are per ticket
        wave_sync();
        infer_step_wave<NC>(c.pack, c.obs, c.acts, c.logp, c.n_agents, t, wmem, env0, n_valid, D
#ifdef RLG_TICK_PROFILE
                            , nullptr
#endif
                            );
        float rew[NC]; bool dn = false;
        if (env_lane) {
            int32_t acts[NC];
            for (int k = 0; k < NC; k++) acts[k] = act_lds[lane * NC + k];
            gym_step_begin<NC>(S.A, S.G, d.cfg, d.action_table, acts);
        }
        wave_sync();
        TickEvents ev; ev.bump_mask = 0;
        arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, seed, env0, ev);
        float* const obs_next = c.obs + ((size_t)(t + 1) * N + (size_t)env * NC) * D;
        if (env_lane) dn = gym_step_after_first_tick<NC>(S.A, S.G, d.cfg, ev, d.action_table, (uint32_t)env, rew, obs_next, (size_t)D, snap);
        if (d.step_stats && env_lane) step_stats_add<NC>(stats, snap);
        if (d.rec_ring && env_lane) write_step_record<NC>(S.A, S.G, snap.touched, dn, d.rec_ring + ((size_t)t * d.n_envs + env) * RLGPU_STEP_RECORD_WORDS(NC));
        wave_sync();
        for (int k = 1; k < d.cfg.tick_skip; k++) { TickEvents ev2; ev2.bump_mask = 0; arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, seed, env0, ev2); }
        if (env_lane) {
            gym_step_end<NC>(S.A, S.G, d.cfg, (uint32_t)env, obs_next, (size_t)D, dn, snap);
            for (int k = 0; k < NC; k++) { c.rew[(size_t)t * N + (size_t)env * NC + k] = rew[k]; c.done[(size_t)t * N + (size_t)env * NC + k] = dn ? 1 : 0; }
            if (RLG_UNLIKELY(dn && d.rec_resets != nullptr)) record_reset<NC>(d.rec_resets, d.rec_count, S.A, S.G, env, t);
            if (c.steps_out) c.steps_out[env] = t + 1;
        }
        wave_sync();
        store_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        wave_sync();
        if (lane == 0) __hip_atomic_store(&c.q_done[g], t + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (d.step_stats) step_stats_flush(d.step_stats, stats, lane);
}

template <int NC>
__global__ void __launch_bounds__(WAVE) k_env_reset(EnvDev d, int run_setter, float* obs, const int32_t* env_ids, int n_ids) {
    constexpr int LANES = lanes_per_block<NC>();
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    if (threadIdx.x >= LANES) return;
    int env = blockIdx.x * LANES + threadIdx.x;
    if (env_ids) { if (env >= n_ids) return; env = env_ids[env]; }   // rlgpu_env_reset_envs: only the listed envs
    if (env < 0 || env >= d.n_envs) return;
    LaneBlock<NC>& S = *reinterpret_cast<LaneBlock<NC>*>(lane_mem + (size_t)threadIdx.x * lane_stride<NC>());
    load_env(d.words, d.n_envs, env, S.A, S.G);
    const int D = obs_size<NC>(d.cfg);
    gym_reset_env<NC>(S.A, S.G, d.cfg, (uint32_t)env, obs ? obs + (size_t)env * players_per_env<NC>(d.cfg) * D : nullptr, (size_t)D, run_setter != 0);
    store_env(d.words, d.n_envs, env, S.A, S.G);
}

// a freshly created env is a fresh arena: every boost pad active (BoostPad's initial state) -- the first Gym::Reset shows the pads as they are
// BEFORE Match::ResetState resets them (arena_gym.h gym_episode_reset), so "all words zero" would show 34 inactive pads
template <int NC>
__global__ void __launch_bounds__(WAVE) k_env_fresh(EnvDev d) {
    constexpr int LANES = lanes_per_block<NC>();
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    if (threadIdx.x >= LANES) return;
    const int env = blockIdx.x * LANES + threadIdx.x;
    if (env >= d.n_envs) return;
    LaneBlock<NC>& S = *reinterpret_cast<LaneBlock<NC>*>(lane_mem + (size_t)threadIdx.x * lane_stride<NC>());
    load_env(d.words, d.n_envs, env, S.A, S.G);
    reset_pads(S.A);
    S.A.mut = mutators_default();
    store_env(d.words, d.n_envs, env, S.A, S.G);
}
// rlgpu_env_set_mutators: every env of the batch gets these MutatorConfig scalars (Arena::SetMutatorConfig on each of the reference's arenas, Gym.cpp:40-44)
template <int NC>
__global__ void __launch_bounds__(WAVE) k_set_mutators(EnvDev d, Mutators m) {
    constexpr int LANES = lanes_per_block<NC>();
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    if (threadIdx.x >= LANES) return;
    const int env = blockIdx.x * LANES + threadIdx.x;
    if (env >= d.n_envs) return;
    LaneBlock<NC>& S = *reinterpret_cast<LaneBlock<NC>*>(lane_mem + (size_t)threadIdx.x * lane_stride<NC>());
    load_env(d.words, d.n_envs, env, S.A, S.G);
    S.A.mut = m;
    store_env(d.words, d.n_envs, env, S.A, S.G);
}

// physics only (rlgpu_env_physics_ticks); with `stamps` (diagnostics, rlgpu_env_debug_tick_cycles) also per workgroup the shader
// cycles (s_memtime) and 100 MHz real-time ticks (s_memrealtime) spent in the tick loop, plus the RLG_TICK_PROFILE phase buckets
template <int NC>
__global__ RLG_NO_TAIL_MARK void __launch_bounds__(WAVE * WPB, RLG_WAVES_PER_SIMD) k_env_ticks(EnvDev d, int ticks, unsigned long long* stamps) {
    constexpr int LANES = lanes_per_block<NC>();
    __shared__ __attribute__((aligned(16))) unsigned char lane_mem[LANES * lane_stride<NC>()];
    __shared__ BvhNode lds_nodes[staged_nodes<NC>()];
    __shared__ uint32_t lds_grid[GRID_WORDS];
    __shared__ uint32_t lds_pad[PAD_TAB_WORDS];
    MeshView mv = stage_mesh(d, lds_nodes, staged_nodes<NC>(), lds_grid, lds_pad);
    const WaveSlot ws = wave_slot<NC>(lane_mem, d.n_envs);
    unsigned char* const wmem = ws.mem; const int env0 = ws.env0, n_valid = ws.n_valid;
    const bool env_lane = ws.lane < n_valid;
    const int env = env0 + (env_lane ? ws.lane : 0);
    LaneBlock<NC>& S = lane_block<NC>(wmem, env_lane ? ws.lane : 0);
#ifdef RLG_POISON_LDS   /* test build: every env byte of the wavefront starts as 0xFF (NaN floats, -1 ints): results must not depend on what LDS held before */
    for (int i = (threadIdx.x & 63); i < (int)((lanes_per_block<NC>() / WPB) * lane_stride<NC>() / 4); i += 64) reinterpret_cast<uint32_t*>(wmem)[i] = 0xFFFFFFFFu;
    wave_sync();
#endif
    epa_arenas_setup<NC>(d, wmem);
    load_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, ws.lane);
    if (env_lane) { S.C.valid = 0; S.C.active = 0; S.C.alive = 0; S.C.watch = 0; }   // candidate lists are per launch
#ifdef RLG_TICK_PROFILE
    if (threadIdx.x == 0) { for (int i = 0; i < 8; i++) g_prof[i] = 0; g_prof_last = __builtin_amdgcn_s_memtime(); }
#endif
#ifdef RLG_FINE_PROF
    if (threadIdx.x == 0) for (int i = 0; i < 64; i++) g_fine[i] = 0;
#endif
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int t = 0; t < ticks; t++) { TickEvents ev; ev.bump_mask = 0; arena_tick_wave<NC>(wmem, n_valid, mv, lds_pad, tick_seed(d.cfg), env0, ev); }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    store_envs_wave<NC>(d.words, d.n_envs, wmem, env0, n_valid, ws.lane);
#ifdef RLG_FINE_PROF
    if (threadIdx.x == 0) for (int i = 0; i < 64; i++) atomicAdd(&g_dbg[i], (int)(g_fine[i] >> 10));   // summed over workgroups, cycles / 1024
    if (threadIdx.x == 0 && blockIdx.x < 4096) for (int i = 0; i < 32; i++) g_fine_blk[32 * blockIdx.x + i] = (unsigned int)(g_fine[i] >> 4);
#endif
    if (stamps && threadIdx.x == 0) {
        unsigned long long* o = stamps + 10 * (size_t)blockIdx.x;  // wave 0 of the workgroup reports
        o[0] = c1 - c0; o[1] = r1 - r0;
#ifdef RLG_TICK_PROFILE
        for (int i = 0; i < 8; i++) o[2 + i] = g_prof[i];
#else
        for (int i = 0; i < 8; i++) o[2 + i] = 0;
#endif
    }
}

__global__ void k_put_infer_pack(rlinfer::InferPack p, rlinfer::InferPack* dst) {
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&p); uint32_t* o = reinterpret_cast<uint32_t*>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(rlinfer::InferPack) / 4); i += blockDim.x) o[i] = s[i];
}

template <int NC>
int env_grid(int n_envs) { return (n_envs + lanes_per_block<NC>() - 1) / lanes_per_block<NC>(); }

// AoS <-> SoA movers for the host fallback path.  An env's working copy (Arena + GymEnv, 1.5 - 3.7 KB) lives in LDS while it is converted: as per-lane
// locals the two structs plus the exchange struct's fields went to scratch memory (k_set_controls<6>: 368 spilled registers; these run every step when a
// plugin is on the host).  MOVE_LANES envs per 64-thread workgroup.
constexpr int MOVE_LANES = 8;
template <int NC> struct MoveBlock { Arena<NC> A; GymEnv<NC> G; };
template <int NC> constexpr size_t move_stride() { size_t w = (sizeof(MoveBlock<NC>) + 7) / 8; return ((w % 2) ? w : w + 1) * 8; }
template <int NC>
__global__ void __launch_bounds__(WAVE) k_upload(EnvDev d, const RlgpuArenaState* src, const int32_t* env_ids, int n) {
    __shared__ __attribute__((aligned(16))) unsigned char mem[MOVE_LANES * move_stride<NC>()];
    if (threadIdx.x >= MOVE_LANES) return;
    int i = blockIdx.x * MOVE_LANES + threadIdx.x;
    if (i >= n) return;
    int env = env_ids ? env_ids[i] : i;
    if (env < 0 || env >= d.n_envs) return;
    // an upload is Arena::SetState on the env's arena, not a new arena: what the broadphase remembers of its proxies stays (bp_hist; all
    // zero in an env that has never ticked = a fresh arena)
    MoveBlock<NC>& S = *reinterpret_cast<MoveBlock<NC>*>(mem + (size_t)threadIdx.x * move_stride<NC>());
    Arena<NC>& A = S.A; GymEnv<NC>& G = S.G;
    load_env(d.words, d.n_envs, env, A, G);
    uint16_t hist[NC + 1];
    for (int b = 0; b <= NC; b++) hist[b] = A.bp_hist[b];
    const uint32_t engine = A.ref_engine;
    const Mutators mut = A.mut;
    arena_from_host(A, G, src[i]);
    if (!(src[i].hidden.valid & RLGPU_HIDDEN_MUTATORS)) A.mut = mut;                                          // (the env keeps the mutators it runs under)
    if (!(src[i].hidden.valid & RLGPU_HIDDEN_BP_HIST)) for (int b = 0; b <= NC; b++) A.bp_hist[b] = hist[b];   // (a state that carries a history brings its own)
    if (!(src[i].hidden.valid & RLGPU_HIDDEN_REF_ENGINE)) A.ref_engine = engine;                              // (likewise the reference-engine test mode: SetState does not touch the thread's engine)
    store_env(d.words, d.n_envs, env, A, G);
}
template <int NC>
__global__ void __launch_bounds__(WAVE) k_download(EnvDev d, RlgpuArenaState* dst, const int32_t* env_ids, int n) {
    __shared__ __attribute__((aligned(16))) unsigned char mem[MOVE_LANES * move_stride<NC>()];
    if (threadIdx.x >= MOVE_LANES) return;
    int i = blockIdx.x * MOVE_LANES + threadIdx.x;
    if (i >= n) return;
    int env = env_ids ? env_ids[i] : i;
    if (env < 0 || env >= d.n_envs) return;
    MoveBlock<NC>& S = *reinterpret_cast<MoveBlock<NC>*>(mem + (size_t)threadIdx.x * move_stride<NC>());
    load_env(d.words, d.n_envs, env, S.A, S.G);
    arena_to_host(S.A, S.G, dst[i]);
}

// Car::controls of every car of every env (Arena facade: car->controls = ...), nothing else of the state touched: the resident words are
// the stepper's own units, so unlike a download / upload pair this does not round anything
template <int NC>
__global__ void __launch_bounds__(WAVE) k_set_controls(EnvDev d, const float* ctl /*[n_envs][NC][8]*/) {
    __shared__ __attribute__((aligned(16))) unsigned char mem[MOVE_LANES * move_stride<NC>()];
    if (threadIdx.x >= MOVE_LANES) return;
    int env = blockIdx.x * MOVE_LANES + threadIdx.x;
    if (env >= d.n_envs) return;
    MoveBlock<NC>& S = *reinterpret_cast<MoveBlock<NC>*>(mem + (size_t)threadIdx.x * move_stride<NC>());
    load_env(d.words, d.n_envs, env, S.A, S.G);
    for (int k = 0; k < NC; k++) S.A.cars[k].ctl = ctl_from(ctl + ((size_t)env * NC + k) * 8);
    store_env(d.words, d.n_envs, env, S.A, S.G);
}

template <int NC>
size_t count_words() {
    Arena<NC> A; GymEnv<NC> G; memset(&A, 0, sizeof(A)); memset(&G, 0, sizeof(G));
    WordCounter c; arena_visit(A, G, c);
    return c.idx;
}

}  // namespace

struct rlgpu_env {
    int device = 0, n_envs = 0, team_size = 1, nc = 2;
    size_t n_words = 0;
    // BallState::rotMat per env (forward / right / up columns).  Under ArenaConfig::noBallRot (ArenaConfig.h:33, the default) the reference never
    // integrates the ball's orientation: it is whatever the last SetState gave it and every GetState hands it back (Ball.cpp:27-49).  The device
    // kernels step with the identity basis (csrc/arena_io.h arena_finish_load) and never see this: uploads store it, downloads report it, an explicit
    // reset by a built-in setter puts the identity back (the resets inside the step kernels leave it alone).
    EnvDev d{};
    BvhNode* d_nodes = nullptr; MeshTri* d_tris = nullptr; float* d_actions = nullptr; uint32_t* d_grid = nullptr; uint32_t* d_pad_tab = nullptr;
    int32_t* d_iota = nullptr;   // 0..n_agents-1 (rlgpu_env_step_controls)
    rlinfer::InferPack* d_infer_pack = nullptr;   // CollectArgs::pack
    int rec_t_cap = 0;                            // steps the step-record ring has room for (rlgpu_env_enable_step_records)
    unsigned int* d_queue = nullptr; int queue_groups = 0, queue_capacity = -1, queue_mode = -1;   // k_env_collect_q: [0] ticket, [16 ..] finished steps per group; -1 auto, 0 never, 1 always
    unsigned int* d_free_counter = nullptr; int free_capacity = -1;   // rlgpu_collect_free: the launch's agent-step counter; workgroups the device keeps resident at once
    unsigned char* d_epa_big = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool timing_on = false;   // rlgpu_env_enable_timing: the step / collect launches are bracketed by events only when asked to (bench, profiling tools)
    // accumulated step-kernel timing: pairs of events recorded around every launch, summed lazily
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool; size_t ev_used = 0; double acc_ms = 0; int acc_launches = 0;
    std::string err;
    struct Redzone { void* base; size_t bytes; const char* name; };
    std::vector<Redzone> redzones;   // RLGPU_REDZONE: the guarded tail of every persistent device buffer (rlgpu_env_check_redzones)
    size_t redzone_bytes = 0;
};

// what-if runs: RLGPU_EXPERIMENT_DYN_LDS=<bytes> of unused dynamic LDS per 1v1 workgroup lowers the workgroups a CU holds (tools/fine_prof.py)
static size_t experiment_dyn_lds() { static const size_t v = [] { const char* s = RLGPU_EXPERIMENT_ENV("RLGPU_EXPERIMENT_DYN_LDS"); return s ? (size_t)atol(s) : (size_t)0; }(); return v; }
// -DRLG_ONLY_NC2: experiment builds (tools/build_variant.sh) instantiate the 1v1 kernels only -- a third of the compile time; rlgpu_env_create refuses other team sizes
#ifdef RLG_ONLY_NC2
#define RLG_NC_PICK(nc, X2, X4, X6) (X2)
#define DISPATCH_NC(e, KERNEL, grid, block, ...) hipLaunchKernelGGL((KERNEL<2>), grid, block, experiment_dyn_lds(), (e)->stream, __VA_ARGS__)
#else
#define RLG_NC_PICK(nc, X2, X4, X6) ((nc) == 2 ? (X2) : ((nc) == 4 ? (X4) : (X6)))
#define DISPATCH_NC(e, KERNEL, grid, block, ...)                                                             \
    do {                                                                                                     \
        if ((e)->nc == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, experiment_dyn_lds(), (e)->stream, __VA_ARGS__);          \
        else if ((e)->nc == 4) hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, (e)->stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL((KERNEL<6>), grid, block, 0, (e)->stream, __VA_ARGS__);                       \
    } while (0)
#endif
#define HIPCHK(e, call)                                                                          \
    do {                                                                                         \
        hipError_t _s = (call);                                                                  \
        if (_s != hipSuccess) {                                                                  \
            (e)->err = std::string(#call) + ": " + hipGetErrorString(_s);                        \
            return RLGPU_ERR_HIP;                                                                \
        }                                                                                        \
    } while (0)

extern "C" {

void rlgpu_default_gym_config(RlgpuGymConfig* c) {
    memset(c, 0, sizeof(*c));
    c->tick_skip = 8;
    c->n_terms = 4;
    c->terms[0] = {RLGPU_RW_FACE_BALL, 0.1f, 0.f};
    c->terms[1] = {RLGPU_RW_VEL_PLAYER_TO_BALL, 0.5f, 0.f};
    c->terms[2] = {RLGPU_RW_VEL_BALL_TO_GOAL, 1.0f, 0.f};
    c->terms[3] = {RLGPU_RW_EVENT, 50.f, 0.f};
    c->event_weights[1] = 1.f;   // teamGoal
    c->event_weights[2] = -1.f;  // concede
    c->zero_sum = 0; c->team_spirit = 0.f; c->opp_scale = 1.f;
    c->n_conds = 2; c->conds[0] = RLGPU_TC_NO_TOUCH; c->conds[1] = RLGPU_TC_GOAL_SCORE; c->no_touch_max_steps = 150;
    c->setter_kind = RLGPU_SS_RANDOM; c->rand_ball_speed = 1; c->rand_car_speed = 1; c->cars_on_ground = 1;
    c->seed_lo = 123; c->seed_hi = 0;
    c->pos_coef[0] = 1 / 4096.f; c->pos_coef[1] = 1 / 5120.f; c->pos_coef[2] = 1 / 2044.f;
    c->vel_coef = 1 / 2300.f; c->ang_vel_coef = 1 / 5.5f;
    c->n_actions = 90;
    c->obs_max_players = 0;
    c->one_team = 0;
}

int rlgpu_pad_location(int pad, float* pos_uu, int* is_big) {
    if (pad < 0 || pad >= RLGPU_NUM_PADS || !pos_uu) return RLGPU_ERR_ARG;
    V3 p = pad_pos(pad);
    pos_uu[0] = p.x; pos_uu[1] = p.y; pos_uu[2] = p.z;
    if (is_big) *is_big = pad < 6 ? 1 : 0;
    return RLGPU_OK;
}

int rlgpu_procedural_mesh(float* verts, int cap_verts, int32_t* tris, int cap_tris, int* n_verts, int* n_tris) {
    std::vector<float> v; std::vector<int32_t> t;
    make_procedural_soccar(v, t);
    *n_verts = (int)v.size() / 3; *n_tris = (int)t.size() / 3;
    if (!verts || !tris) return RLGPU_OK;
    if (*n_verts > cap_verts || *n_tris > cap_tris) return RLGPU_ERR_ARG;
    memcpy(verts, v.data(), v.size() * 4); memcpy(tris, t.data(), t.size() * 4);
    return RLGPU_OK;
}

int rlgpu_procedural_mesh_ex(int fillet_segments, float max_edge_uu, float* verts, int cap_verts, int32_t* tris, int cap_tris, int* n_verts, int* n_tris) {
    std::vector<float> v; std::vector<int32_t> t;
    make_procedural_soccar_ex(v, t, fillet_segments, max_edge_uu);
    *n_verts = (int)v.size() / 3; *n_tris = (int)t.size() / 3;
    if (!verts || !tris) return RLGPU_OK;
    if (*n_verts > cap_verts || *n_tris > cap_tris) return RLGPU_ERR_ARG;
    memcpy(verts, v.data(), v.size() * 4); memcpy(tris, t.data(), t.size() * 4);
    return RLGPU_OK;
}

int rlgpu_action_table(float* out, int cap_rows) {
    float tab[90 * 8];
    int n = build_action_table(tab);
    if (out) memcpy(out, tab, sizeof(float) * 8 * (size_t)std::min(n, cap_rows));
    return n;
}

// Debug mode RLGPU_REDZONE=<bytes> (read at create): every persistent device buffer of the env batch gets that many extra bytes behind it, filled
// with 0xC5; rlgpu_env_check_redzones reads them back and names the first buffer something wrote past.  (Round 4's staged word rows ran NC
// rows past the resident words and were found only when they happened to cross into unmapped memory; with this mode a test sees such a store
// wherever the allocator put the buffer.)
static hipError_t rz_malloc_raw(rlgpu_env* e, void** p, size_t bytes, const char* name) {
    const size_t rz = e->redzone_bytes;
    hipError_t r = hipMalloc(p, bytes + rz);
    if (r != hipSuccess || !rz) return r;
    r = hipMemset((char*)*p + bytes, 0xC5, rz);
    e->redzones.push_back({*p, bytes, name});
    return r;
}
#define RZ_MALLOC(e, ptr, bytes, name) rz_malloc_raw(e, (void**)&(ptr), bytes, name)
static void rz_free(rlgpu_env* e, void* p) {
    for (size_t i = 0; i < e->redzones.size(); i++) if (e->redzones[i].base == p) { e->redzones.erase(e->redzones.begin() + (long)i); break; }
    (void)hipFree(p);
}
// test hook of the checker itself: one byte written `past` bytes behind guarded buffer number `which` (creation order), as a stray store would
int rlgpu_env_debug_overrun(rlgpu_env* e, int which, int past) {
    if (!e->redzone_bytes || which < 0 || which >= (int)e->redzones.size() || past < 0 || (size_t)past >= e->redzone_bytes) { e->err = "rlgpu_env_debug_overrun: no such guard byte"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemset((char*)e->redzones[(size_t)which].base + e->redzones[(size_t)which].bytes + past, 0, 1));
    return RLGPU_OK;
}
int rlgpu_env_check_redzones(rlgpu_env* e) {
    if (!e->redzone_bytes) { e->err = "rlgpu_env_check_redzones: the batch was created without RLGPU_REDZONE"; return RLGPU_ERR_STATE; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipDeviceSynchronize());
    std::vector<unsigned char> h(e->redzone_bytes);
    for (const auto& z : e->redzones) {
        HIPCHK(e, hipMemcpy(h.data(), (const char*)z.base + z.bytes, h.size(), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < h.size(); i++) if (h[i] != 0xC5) {
            size_t last = i; for (size_t j = i; j < h.size(); j++) if (h[j] != 0xC5) last = j;
            e->err = std::string("redzone of '") + z.name + "' (" + std::to_string(z.bytes) + " bytes) overwritten: first at +" + std::to_string(i) + ", last at +" + std::to_string(last) + " past its end";
            return RLGPU_ERR_STATE;
        }
    }
    return RLGPU_OK;
}

int rlgpu_env_create(rlgpu_env** out, int device, int n_envs, int team_size, const RlgpuGymConfig* cfg) {
    if (!out || n_envs <= 0 || team_size < 1 || team_size > 3 || !cfg) return RLGPU_ERR_ARG;
    if (cfg->obs_max_players != 0 && (cfg->obs_max_players < team_size || cfg->obs_max_players > rlg::OBS_MAX_PADDED_PLAYERS)) return RLGPU_ERR_ARG;   // DefaultOBSPadded.cpp:40-44: too many players for the padding
#ifdef RLG_ONLY_NC2
    if (team_size != 1) return RLGPU_ERR_ARG;   // (an experiment build: 1v1 kernels only)
#endif
    rlgpu_env* e = new rlgpu_env();
    *out = e;
    e->device = device; e->n_envs = n_envs; e->team_size = team_size; e->nc = 2 * team_size;
    HIPCHK(e, hipSetDevice(device));
    { const char* rz = getenv("RLGPU_REDZONE"); e->redzone_bytes = rz ? (size_t)atol(rz) : 0; }
    e->n_words = RLG_NC_PICK(e->nc, count_words<2>(), count_words<4>(), count_words<6>());
    {   // the kernels stage arena_num_words<NC>() rows, the visitor defines how many there are: one number, or nothing runs
        const size_t staged = RLG_NC_PICK(e->nc, arena_num_words<2>(), arena_num_words<4>(), arena_num_words<6>());
#ifndef RLG_TEST_EXTRA_WORD_ROWS
        if (staged != e->n_words) { e->err = "arena_num_words disagrees with arena_visit (" + std::to_string(staged) + " vs " + std::to_string(e->n_words) + " words per env)"; return RLGPU_ERR_ARG; }
#else
        (void)staged;
#endif
    }
    HIPCHK(e, RZ_MALLOC(e, e->d.words, e->n_words * (size_t)n_envs * 4, "resident state words"));
    HIPCHK(e, hipMemset(e->d.words, 0, e->n_words * (size_t)n_envs * 4));
    float tab[90 * 8]; build_action_table(tab);
    HIPCHK(e, RZ_MALLOC(e, e->d_actions, sizeof(tab), "action table"));
    HIPCHK(e, hipMemcpy(e->d_actions, tab, sizeof(tab), hipMemcpyHostToDevice));
    e->d.action_table = e->d_actions;
    uint32_t ptab[PAD_TAB_WORDS]; pad_table_fill(ptab);
    HIPCHK(e, RZ_MALLOC(e, e->d_pad_tab, sizeof(ptab), "boost pad table"));
    HIPCHK(e, hipMemcpy(e->d_pad_tab, ptab, sizeof(ptab), hipMemcpyHostToDevice));
    e->d.pad_tab = e->d_pad_tab;
    memcpy(&e->d.cfg, cfg, sizeof(GymConfig));
    {   // full-size penetration-depth arenas, one per wavefront of a step launch (11.7 KB each; touched only by the rare query the LDS arena cannot hold)
        const size_t waves = (size_t)(RLG_NC_PICK(e->nc, env_grid<2>(n_envs), env_grid<4>(n_envs), env_grid<6>(n_envs))) * WPB;
        HIPCHK(e, RZ_MALLOC(e, e->d_epa_big, waves * EPA_BIG_BYTES, "EPA arenas"));
        e->d.epa_big = e->d_epa_big;
        HIPCHK(e, RZ_MALLOC(e, e->d.leaf_cache, (size_t)n_envs * (e->nc + 1) * CACHE_LEAVES * sizeof(uint32_t), "candidate leaf cache"));   // CandCache: 0.4 - 0.9 KB per env
        {   // the big contact layout's pool (tick_world_big): 64 slots of 0.1 - 0.3 MB
            const size_t slot = RLG_NC_PICK(e->nc, big_work_bytes<2>(), big_work_bytes<4>(), big_work_bytes<6>());
            HIPCHK(e, RZ_MALLOC(e, e->d.big_work, slot * BIG_WORK_SLOTS, "big contact layout pool"));
            HIPCHK(e, RZ_MALLOC(e, e->d.big_locks, sizeof(uint32_t) * BIG_WORK_SLOTS, "big contact layout locks"));
            HIPCHK(e, hipMemset(e->d.big_work, 0, slot * BIG_WORK_SLOTS));
            HIPCHK(e, hipMemset(e->d.big_locks, 0, sizeof(uint32_t) * BIG_WORK_SLOTS));
        }
        // Both scratch buffers start as zeros (RLGPU_SCRATCH_FILL=<byte>: another pattern, for tests): hipMalloc hands back whatever an earlier
        // allocation of the process left there; nothing reads either before writing it, and a known start makes that checkable
        {
            const char* f = RLGPU_EXPERIMENT_ENV("RLGPU_SCRATCH_FILL");
            const int fill = f ? (int)std::strtol(f, nullptr, 0) : 0;
            HIPCHK(e, hipMemset(e->d_epa_big, fill, waves * EPA_BIG_BYTES));
            HIPCHK(e, hipMemset(e->d.leaf_cache, fill, (size_t)n_envs * (e->nc + 1) * CACHE_LEAVES * sizeof(uint32_t)));
        }
    }
    e->d.n_envs = n_envs; e->d.nodes = nullptr; e->d.tris = nullptr; e->d.n_nodes = 0; e->d.n_tris = 0; e->d.grid = nullptr; e->d.cand_fat = RLG_CAND_FAT;
    {
        dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(n_envs), env_grid<4>(n_envs), env_grid<6>(n_envs))), block(WAVE);
        DISPATCH_NC(e, k_env_fresh, grid, block, e->d);
        HIPCHK(e, hipGetLastError());
        HIPCHK(e, hipDeviceSynchronize());
    }
    return RLGPU_OK;
}

void rlgpu_env_destroy(rlgpu_env* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->d.words) (void)hipFree(e->d.words);
    if (e->d_nodes) (void)hipFree(e->d_nodes);
    if (e->d_tris) (void)hipFree(e->d_tris);
    if (e->d_actions) (void)hipFree(e->d_actions);
    if (e->d_grid) (void)hipFree(e->d_grid);
    if (e->d_pad_tab) (void)hipFree(e->d_pad_tab);
    if (e->d.snap_out) (void)hipFree(e->d.snap_out);
    if (e->d_iota) (void)hipFree(e->d_iota);
    if (e->d_infer_pack) (void)hipFree(e->d_infer_pack);
    if (e->d.rec_ring) (void)hipFree(e->d.rec_ring);
    if (e->d.rec_resets) (void)hipFree(e->d.rec_resets);
    if (e->d.rec_count) (void)hipFree(e->d.rec_count);
    if (e->d_free_counter) (void)hipFree(e->d_free_counter);
    if (e->d_queue) (void)hipFree(e->d_queue);
    if (e->d_epa_big) (void)hipFree(e->d_epa_big);
    if (e->d.leaf_cache) (void)hipFree(e->d.leaf_cache);
    if (e->d.big_work) (void)hipFree(e->d.big_work);
    if (e->d.big_locks) (void)hipFree(e->d.big_locks);
    if (e->d.step_stats) (void)hipFree(e->d.step_stats);
    for (auto& p : e->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    delete e;
}
const char* rlgpu_env_last_error(const rlgpu_env* e) { return e ? e->err.c_str() : "null env"; }
int rlgpu_env_reseed(rlgpu_env* e, uint32_t seed_lo, uint32_t seed_hi) { e->d.cfg.seed_lo = seed_lo; e->d.cfg.seed_hi = seed_hi; return RLGPU_OK; }
int rlgpu_env_set_stream(rlgpu_env* e, void* s) { e->stream = (hipStream_t)s; return RLGPU_OK; }
int rlgpu_env_obs_size(const rlgpu_env* e) { return e->d.cfg.obs_max_players > 0 ? 51 + 38 * e->d.cfg.obs_max_players : 51 + 19 * (e->d.cfg.one_team ? e->nc / 2 : e->nc); }
int rlgpu_env_num_agents(const rlgpu_env* e) { return e->n_envs * (e->d.cfg.one_team ? e->nc / 2 : e->nc); }
int rlgpu_env_num_actions(const rlgpu_env* e) { return e->d.cfg.n_actions; }
int rlgpu_env_state_words(const rlgpu_env* e) { return (int)e->n_words; }
int rlgpu_state_word_counts(int team_size, int* visited, int* staged) {
    if (team_size < 1 || team_size > 3 || !visited || !staged) return RLGPU_ERR_ARG;
    *visited = (int)(RLG_NC_PICK(2 * team_size, count_words<2>(), count_words<4>(), count_words<6>()));
    *staged = (int)(RLG_NC_PICK(2 * team_size, arena_num_words<2>(), arena_num_words<4>(), arena_num_words<6>()));
    return RLGPU_OK;
}

static int env_set_mesh_parts(rlgpu_env* e, const float* verts, int n_verts, const int32_t* tris, int n_tris, const std::vector<int>* parts, bool verts_in_bt = false);
int rlgpu_env_set_mesh(rlgpu_env* e, const float* verts, int n_verts, const int32_t* tris, int n_tris) { return env_set_mesh_parts(e, verts, n_verts, tris, n_tris, nullptr); }
int rlgpu_mesh_visit_order(const float* verts, int n_verts, const int32_t* tris, int n_tris, int32_t* order_out) {
    if (!verts || !tris || !order_out || n_tris < 0) return RLGPU_ERR_ARG;
    HostMesh m = build_host_mesh(verts, n_verts, tris, n_tris);
    for (int i = 0; i < n_tris; i++) order_out[i] = m.source_tri[i];
    return RLGPU_OK;
}
static int env_set_mesh_parts(rlgpu_env* e, const float* verts, int n_verts, const int32_t* tris, int n_tris, const std::vector<int>* parts, bool verts_in_bt) {
    HIPCHK(e, hipSetDevice(e->device));
    HostMesh m = build_host_mesh(verts, n_verts, tris, n_tris, parts, verts_in_bt);
    if (e->d_nodes) { rz_free(e, e->d_nodes); e->d_nodes = nullptr; }
    if (e->d_tris) { rz_free(e, e->d_tris); e->d_tris = nullptr; }
    e->d.n_nodes = (int)m.nodes.size(); e->d.n_tris = (int)m.tris.size();
    e->d.cand_fat = e->d.n_tris > RLG_CAND_DENSE_TRIS ? RLG_CAND_FAT_DENSE : RLG_CAND_FAT;
    if (!m.nodes.empty()) {
        HIPCHK(e, RZ_MALLOC(e, e->d_nodes, m.nodes.size() * sizeof(BvhNode), "BVH nodes"));
        HIPCHK(e, RZ_MALLOC(e, e->d_tris, m.tris.size() * sizeof(MeshTri), "mesh triangles"));
        HIPCHK(e, hipMemcpy(e->d_nodes, m.nodes.data(), m.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->d_tris, m.tris.data(), m.tris.size() * sizeof(MeshTri), hipMemcpyHostToDevice));
    }
    if (e->d_grid) { rz_free(e, e->d_grid); e->d_grid = nullptr; }
    if (!m.grid.empty()) {
        HIPCHK(e, RZ_MALLOC(e, e->d_grid, m.grid.size() * 4, "occupancy grid"));
        HIPCHK(e, hipMemcpy(e->d_grid, m.grid.data(), m.grid.size() * 4, hipMemcpyHostToDevice));
    }
    e->d.nodes = e->d_nodes; e->d.tris = e->d_tris; e->d.grid = e->d_grid;
    return RLGPU_OK;
}
int rlgpu_env_set_procedural_mesh(rlgpu_env* e) {
    std::vector<float> v; std::vector<int32_t> t;
    make_procedural_soccar(v, t);
    return rlgpu_env_set_mesh(e, v.data(), (int)v.size() / 3, t.data(), (int)t.size() / 3);
}
int rlgpu_env_load_cmf_dir(rlgpu_env* e, const char* dir) {
    DIR* dp = opendir(dir);
    if (!dp) { e->err = std::string("cannot open ") + dir; return RLGPU_ERR_ARG; }
    std::vector<std::string> files;
    while (dirent* de = readdir(dp)) { std::string n = de->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".cmf") files.push_back(std::string(dir) + "/" + n); }
    closedir(dp);
    std::sort(files.begin(), files.end());
    if (files.empty()) { e->err = std::string("no .cmf files in ") + dir; return RLGPU_ERR_ARG; }
    std::vector<float> v; std::vector<int32_t> t;
    std::vector<int> parts;   // one mesh object per file, as in the reference (RocketSim.cpp:149-167): each keeps its own triangle order
    for (auto& f : files) {
        std::ifstream in(f, std::ios::binary);
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        const size_t before = t.size() / 3;
        if (!append_cmf(buf.data(), buf.size(), v, t, true)) { e->err = "bad cmf file " + f; return RLGPU_ERR_ARG; }   // vertices stay in Bullet units, bit for bit
        parts.push_back((int)(t.size() / 3 - before));
    }
    return env_set_mesh_parts(e, v.data(), (int)v.size() / 3, t.data(), (int)t.size() / 3, &parts, true);
}



void rlgpu_default_mutators(RlgpuMutators* m) { *m = mutators_to_abi(mutators_default()); }
float rlgpu_ball_damp_per_tick(float ball_drag) { return powf(1.0f - ball_drag, TICK_DT); }   // btRigidBody::applyDamping's factor, by the host's C library
int rlgpu_env_set_mutators(rlgpu_env* e, const RlgpuMutators* m) {
    if (!m) { e->err = "rlgpu_env_set_mutators: null"; return RLGPU_ERR_ARG; }
    if (!(m->ball_damp_per_tick > 0.f && m->ball_damp_per_tick <= 1.f) || !(m->ball_max_speed >= 0.f) || (m->flags & ~63u) ||
        ((m->flags & RLGPU_MUT_DEMO_ON_CONTACT) && (m->flags & RLGPU_MUT_DEMO_DISABLED))) {
        e->err = "rlgpu_env_set_mutators: ball_damp_per_tick must be in (0, 1] (rlgpu_ball_damp_per_tick(ballDrag)), ball_max_speed >= 0, flags a combination of RLGPU_MUT_* with at most one demo mode";
        return RLGPU_ERR_ARG;
    }
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(e->d.n_envs), env_grid<4>(e->d.n_envs), env_grid<6>(e->d.n_envs))), block(WAVE);
    DISPATCH_NC(e, k_set_mutators, grid, block, e->d, mutators_from_abi(*m));
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return RLGPU_OK;
}
int rlgpu_env_upload_states(rlgpu_env* e, const RlgpuArenaState* host, const int32_t* env_ids, int n) {
    if (n <= 0) return RLGPU_OK;
    HIPCHK(e, hipSetDevice(e->device));
    RlgpuArenaState* dsrc = nullptr; int32_t* dids = nullptr;
    HIPCHK(e, hipMalloc(&dsrc, sizeof(RlgpuArenaState) * (size_t)n));
    HIPCHK(e, hipMemcpyAsync(dsrc, host, sizeof(RlgpuArenaState) * (size_t)n, hipMemcpyHostToDevice, e->stream));
    if (env_ids) { HIPCHK(e, hipMalloc(&dids, 4 * (size_t)n)); HIPCHK(e, hipMemcpyAsync(dids, env_ids, 4 * (size_t)n, hipMemcpyHostToDevice, e->stream)); }
    dim3 grid((n + MOVE_LANES - 1) / MOVE_LANES), block(64);
    DISPATCH_NC(e, k_upload, grid, block, e->d, (const RlgpuArenaState*)dsrc, (const int32_t*)dids, n);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    (void)hipFree(dsrc); if (dids) (void)hipFree(dids);
    return RLGPU_OK;
}
int rlgpu_env_download_states(rlgpu_env* e, RlgpuArenaState* host, const int32_t* env_ids, int n) {
    if (n <= 0) return RLGPU_OK;
    HIPCHK(e, hipSetDevice(e->device));
    RlgpuArenaState* ddst = nullptr; int32_t* dids = nullptr;
    HIPCHK(e, hipMalloc(&ddst, sizeof(RlgpuArenaState) * (size_t)n));
    if (env_ids) { HIPCHK(e, hipMalloc(&dids, 4 * (size_t)n)); HIPCHK(e, hipMemcpyAsync(dids, env_ids, 4 * (size_t)n, hipMemcpyHostToDevice, e->stream)); }
    // every byte of what the caller gets is defined: car slots beyond the env's cars, the reserved part of the appended block and the struct's
    // padding are zeros (arena_to_host writes the live fields only)
    HIPCHK(e, hipMemsetAsync(ddst, 0, sizeof(RlgpuArenaState) * (size_t)n, e->stream));
    dim3 grid((n + MOVE_LANES - 1) / MOVE_LANES), block(64);
    DISPATCH_NC(e, k_download, grid, block, e->d, ddst, (const int32_t*)dids, n);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(host, ddst, sizeof(RlgpuArenaState) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    (void)hipFree(ddst); if (dids) (void)hipFree(dids);
    return RLGPU_OK;
}

int rlgpu_env_reset(rlgpu_env* e, int run_setter, float* obs_dev) {
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(e->n_envs), env_grid<4>(e->n_envs), env_grid<6>(e->n_envs))), block(WAVE);
    DISPATCH_NC(e, k_env_reset, grid, block, e->d, run_setter, obs_dev, (const int32_t*)nullptr, 0);
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_env_reset_envs(rlgpu_env* e, const int32_t* env_ids, int n, int run_setter, float* obs_dev) {
    if (n <= 0) return RLGPU_OK;
    if (!env_ids) { e->err = "rlgpu_env_reset_envs: null env list"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    int32_t* dids = nullptr;
    HIPCHK(e, hipMalloc(&dids, 4 * (size_t)n));
    HIPCHK(e, hipMemcpyAsync(dids, env_ids, 4 * (size_t)n, hipMemcpyHostToDevice, e->stream));
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(n), env_grid<4>(n), env_grid<6>(n))), block(WAVE);
    DISPATCH_NC(e, k_env_reset, grid, block, e->d, run_setter, obs_dev, (const int32_t*)dids, n);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    (void)hipFree(dids);
    return RLGPU_OK;
}

int rlgpu_env_enable_step_stats(rlgpu_env* e, int on) {
    HIPCHK(e, hipSetDevice(e->device));
    if (on && !e->d.step_stats) {
        HIPCHK(e, RZ_MALLOC(e, e->d.step_stats, 4 * sizeof(float), "step statistics"));
        HIPCHK(e, hipMemsetAsync(e->d.step_stats, 0, 4 * sizeof(float), e->stream));
    } else if (!on && e->d.step_stats) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        rz_free(e, e->d.step_stats); e->d.step_stats = nullptr;
    }
    return RLGPU_OK;
}
int rlgpu_env_step_stats(rlgpu_env* e, float* out4, int reset) {
    if (!e->d.step_stats) { e->err = "rlgpu_env_step_stats: call rlgpu_env_enable_step_stats first"; return RLGPU_ERR_STATE; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(out4, e->d.step_stats, 4 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    if (reset) HIPCHK(e, hipMemsetAsync(e->d.step_stats, 0, 4 * sizeof(float), e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return RLGPU_OK;
}

int rlgpu_env_enable_snapshots(rlgpu_env* e, int on) {
    HIPCHK(e, hipSetDevice(e->device));
    if (on && !e->d.snap_out) {
        HIPCHK(e, RZ_MALLOC(e, e->d.snap_out, sizeof(RlgpuArenaState) * (size_t)e->n_envs, "snapshots"));
        HIPCHK(e, hipMemsetAsync(e->d.snap_out, 0, sizeof(RlgpuArenaState) * (size_t)e->n_envs, e->stream));
    } else if (!on && e->d.snap_out) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        rz_free(e, e->d.snap_out); e->d.snap_out = nullptr;
    }
    return RLGPU_OK;
}

// ---- step records: the GameState source of every step of a collection launch, for plugins that run after it (a user RewardFunction, a step callback) ----
int rlgpu_env_enable_step_records(rlgpu_env* e, int t_cap) {
    HIPCHK(e, hipSetDevice(e->device));
    if (t_cap < 0) { e->err = "rlgpu_env_enable_step_records: t_cap < 0"; return RLGPU_ERR_ARG; }
    if (e->d.rec_ring) { HIPCHK(e, hipStreamSynchronize(e->stream)); rz_free(e, e->d.rec_ring); rz_free(e, e->d.rec_resets); rz_free(e, e->d.rec_count); e->d.rec_ring = e->d.rec_resets = nullptr; e->d.rec_count = nullptr; e->rec_t_cap = 0; }
    if (t_cap == 0) return RLGPU_OK;
    const size_t W = (size_t)RLGPU_STEP_RECORD_WORDS(e->nc), slots = (size_t)t_cap * (size_t)e->n_envs;
    HIPCHK(e, RZ_MALLOC(e, e->d.rec_ring, slots * W * 4, "step records"));
    HIPCHK(e, RZ_MALLOC(e, e->d.rec_resets, slots * (W + 2) * 4, "reset records"));
    HIPCHK(e, RZ_MALLOC(e, e->d.rec_count, 64, "reset record count"));
    HIPCHK(e, hipMemsetAsync(e->d.rec_count, 0, 64, e->stream));
    e->rec_t_cap = t_cap;
    return RLGPU_OK;
}
int rlgpu_env_step_record_words(const rlgpu_env* e) { return RLGPU_STEP_RECORD_WORDS(e->nc); }
int rlgpu_env_download_step_records(rlgpu_env* e, int t_used, uint32_t* host_ring, uint32_t* host_resets, int reset_cap, int* n_resets) {
    if (!e->d.rec_ring) { e->err = "rlgpu_env_download_step_records: call rlgpu_env_enable_step_records first"; return RLGPU_ERR_STATE; }
    if (t_used < 0 || t_used > e->rec_t_cap || !host_ring || !n_resets || (reset_cap > 0 && !host_resets)) { e->err = "rlgpu_env_download_step_records: bad argument"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    const size_t W = (size_t)RLGPU_STEP_RECORD_WORDS(e->nc);
    unsigned int cnt = 0;
    HIPCHK(e, hipMemcpyAsync(&cnt, e->d.rec_count, 4, hipMemcpyDeviceToHost, e->stream));
    if (t_used > 0) HIPCHK(e, hipMemcpyAsync(host_ring, e->d.rec_ring, (size_t)t_used * e->n_envs * W * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    *n_resets = (int)cnt;
    if ((int)cnt > reset_cap) { e->err = "rlgpu_env_download_step_records: more reset records than the caller has room for"; return RLGPU_ERR_ARG; }
    if (cnt) HIPCHK(e, hipMemcpy(host_resets, e->d.rec_resets, (size_t)cnt * (W + 2) * 4, hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemsetAsync(e->d.rec_count, 0, 4, e->stream));   // the next launch appends from the start
    return RLGPU_OK;
}

int rlgpu_env_download_snapshots(rlgpu_env* e, RlgpuArenaState* host, int first_env, int n) {
    if (!e->d.snap_out) { e->err = "rlgpu_env_download_snapshots: call rlgpu_env_enable_snapshots first"; return RLGPU_ERR_STATE; }
    if (first_env < 0 || n < 0 || first_env + n > e->n_envs) { e->err = "rlgpu_env_download_snapshots: env range out of bounds"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipMemcpyAsync(host, e->d.snap_out + first_env, sizeof(RlgpuArenaState) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return RLGPU_OK;
}

// the next pair of timing events of the pool (drained into the running totals when all 2048 are in use)
static int env_next_events(rlgpu_env* e, std::pair<hipEvent_t, hipEvent_t>** out) {
    if (e->ev_used == e->ev_pool.size()) {
        if (e->ev_pool.size() < 2048) {
            hipEvent_t a, b; HIPCHK(e, hipEventCreate(&a)); HIPCHK(e, hipEventCreate(&b));
            e->ev_pool.push_back({a, b});
        } else {
            float tmp; int n; int rc = rlgpu_env_timing_total(e, &tmp, &n, 0);
            if (rc) return rc;
        }
    }
    *out = &e->ev_pool[e->ev_used++];
    return RLGPU_OK;
}

int rlgpu_env_enable_timing(rlgpu_env* e, int on) { e->timing_on = on != 0; if (!on) e->timed = false; return RLGPU_OK; }

int rlgpu_env_step(rlgpu_env* e, const int32_t* actions, float* next_obs, float* reward, int32_t* done) {
    if (!actions || !next_obs || !reward || !done) { e->err = "rlgpu_env_step: null device pointer"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(e->n_envs), env_grid<4>(e->n_envs), env_grid<6>(e->n_envs))), block(WAVE * WPB);
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    if (e->timing_on) { int rc_ev = env_next_events(e, &evp); if (rc_ev) return rc_ev; HIPCHK(e, hipEventRecord(evp->first, e->stream)); }
    DISPATCH_NC(e, k_env_step, grid, block, e->d, actions, next_obs, reward, done);
    if (evp) { HIPCHK(e, hipEventRecord(evp->second, e->stream)); e->ev0 = evp->first; e->ev1 = evp->second; e->timed = true; }
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

// Gym::Step with the controls already parsed on the host (a user ActionParser, a standalone Gym): row r of controls_dev is agent r's
// Action (8 floats).  Same kernel: the "action table" it indexes is the controls buffer itself and every agent's action is its own row.
int rlgpu_env_step_controls(rlgpu_env* e, const float* controls, float* next_obs, float* reward, int32_t* done) {
    if (!controls || !next_obs || !reward || !done) { e->err = "rlgpu_env_step_controls: null device pointer"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    const int n_agents = rlgpu_env_num_agents(e);
    if (!e->d_iota) {
        std::vector<int32_t> iota((size_t)n_agents);
        for (int i = 0; i < n_agents; i++) iota[i] = i;
        HIPCHK(e, RZ_MALLOC(e, e->d_iota, 4 * (size_t)n_agents, "iota"));
        HIPCHK(e, hipMemcpy(e->d_iota, iota.data(), 4 * (size_t)n_agents, hipMemcpyHostToDevice));
    }
    EnvDev d = e->d;
    d.action_table = controls; d.cfg.n_actions = n_agents;
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(e->n_envs), env_grid<4>(e->n_envs), env_grid<6>(e->n_envs))), block(WAVE * WPB);
    DISPATCH_NC(e, k_env_step, grid, block, d, (const int32_t*)e->d_iota, next_obs, reward, done);
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

// ThreadAgent::_RunFunc for a whole collection phase (ThreadAgent.cpp:58-163): T x (policy->GetAction, GameInst::Step) for every env
static int collect_impl(rlgpu_env* e, rlgpu_learner* l, int T, float* obs, int32_t* actions, float* logp, float* reward, int32_t* done, int deterministic,
                        int64_t free_target, int32_t* steps_out, const char* who) {
    if (!l || T <= 0 || !obs || !actions || !logp || !reward || !done) { e->err = std::string(who) + ": bad argument"; return RLGPU_ERR_ARG; }
    if (e->d.cfg.one_team) { e->err = std::string(who) + ": one-team envs are collected step by step (rlgpu_policy_act + rlgpu_env_step)"; return RLGPU_ERR_STATE; }
    HIPCHK(e, hipSetDevice(e->device));
    CollectArgs c{};
    rlinfer::InferPack pk{};
    const int epw = (RLG_NC_PICK(e->nc, lanes_per_block<2>(), lanes_per_block<4>(), lanes_per_block<6>())) / WPB;
    const size_t tw = RLG_NC_PICK(e->nc, sizeof(TickWork<2>), sizeof(TickWork<4>), sizeof(TickWork<6>));
    const int max_buf = (int)((tw - 64) / (epw >= 2 ? 1 : 2));
    const int half_buf = (int)((tw - 64) / (epw >= 4 ? 1 : (epw >= 2 ? 2 : 4))) & ~15;     // fp32 mode: what one part of an activation buffer may take
    int rc = rlgpu_internal_policy_net(l, &pk.net, &pk.head, deterministic, T, -half_buf, (void*)e->stream);
    if (rc == RLGPU_OK && !pk.net.fp32 && rlinfer::wave_buf_bytes(epw * e->nc, pk.net.ld) > max_buf) rc = RLGPU_ERR_STATE;
    if (rc == RLGPU_OK && pk.net.fp32 && rlinfer::f32_part_bytes(epw * e->nc, epw == 3 ? 3 : 2, pk.net.ld) > half_buf) rc = RLGPU_ERR_STATE;
    if (rc) { e->err = std::string(who) + ": the policy does not fit the in-kernel inference (<= 128 actions, hidden width within the LDS scratch)"; return rc; }
    if (pk.net.D != rlgpu_env_obs_size(e)) { e->err = std::string(who) + ": the policy's input width is not the env's observation width"; return RLGPU_ERR_ARG; }
    // the net's description goes to device memory on the env's stream (a kernel with the struct as its argument: stream-ordered with the launches that read it, no pinned staging)
    if (!e->d_infer_pack) HIPCHK(e, RZ_MALLOC(e, e->d_infer_pack, sizeof(rlinfer::InferPack), "policy description of the in-kernel inference"));
    hipLaunchKernelGGL(k_put_infer_pack, dim3(1), dim3(64), 0, e->stream, pk, e->d_infer_pack);
    c.pack = e->d_infer_pack;
    if (e->d.rec_ring && T > e->rec_t_cap) { e->err = std::string(who) + ": more steps than the step-record ring has room for (rlgpu_env_enable_step_records)"; return RLGPU_ERR_ARG; }
    c.T = T; c.n_agents = e->n_envs * e->nc; c.obs = obs; c.acts = actions; c.logp = logp; c.rew = reward; c.done = done;
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(e->n_envs), env_grid<4>(e->n_envs), env_grid<6>(e->n_envs))), block(WAVE * WPB);
    if (free_target > 0) {
        // The agents must all be RUNNING for "stop when the total is there" to mean what it means in the reference (every agent thread runs
        // from the start): all workgroups of the launch resident at once.  A batch with more wavefronts than the device holds is load-balanced
        // by the dispatcher anyway (a retiring wavefront makes room for the next) and is collected in lockstep (RLGPU_ERR_STATE: rlgpu_collect).
        if (e->free_capacity < 0) {
            int per_cu = 0; hipDeviceProp_t prop{};
            HIPCHK(e, hipGetDeviceProperties(&prop, e->device));
            hipError_t oc = RLG_NC_PICK(e->nc, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_env_collect<2>, WAVE * WPB, 0),
                                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_env_collect<4>, WAVE * WPB, 0),
                                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_env_collect<6>, WAVE * WPB, 0));
            HIPCHK(e, oc);
            e->free_capacity = per_cu * prop.multiProcessorCount;
        }
        if ((int)grid.x > e->free_capacity) { e->err = std::string(who) + ": more workgroups than the device keeps resident at once; collect in lockstep (rlgpu_collect)"; return RLGPU_ERR_STATE; }
        if (free_target > 0xFFFFFFFFll - (int64_t)c.n_agents || !steps_out) { e->err = std::string(who) + ": bad target / steps_out"; return RLGPU_ERR_ARG; }
        if (!e->d_free_counter) HIPCHK(e, RZ_MALLOC(e, e->d_free_counter, 64, "free-running counter"));
        HIPCHK(e, hipMemsetAsync(e->d_free_counter, 0, 4, e->stream));
        c.counter = e->d_free_counter; c.free_target = (unsigned int)free_target; c.steps_out = steps_out;
    } else c.steps_out = steps_out;
    // lockstep collection of more wavefront-groups than stay resident: the step queue (k_env_collect_q)
    bool queued = false;
    if (free_target <= 0 && e->queue_mode != 0) {
        if (e->queue_capacity < 0) {
            int per_cu = 0; hipDeviceProp_t prop{};
            HIPCHK(e, hipGetDeviceProperties(&prop, e->device));
            hipError_t oc = RLG_NC_PICK(e->nc, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_env_collect_q<2>, WAVE * WPB, 0),
                                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_env_collect_q<4>, WAVE * WPB, 0),
                                        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_env_collect_q<6>, WAVE * WPB, 0));
            HIPCHK(e, oc);
            e->queue_capacity = per_cu * prop.multiProcessorCount;
        }
        const int groups = (int)grid.x * WPB;
        if (e->queue_capacity > 0 && (e->queue_mode == 1 || (int)grid.x > e->queue_capacity)) {
            if (!e->d_queue || e->queue_groups < groups) {
                if (e->d_queue) rz_free(e, e->d_queue);
                e->d_queue = nullptr;
                HIPCHK(e, RZ_MALLOC(e, e->d_queue, 4 * (size_t)(16 + groups), "collection step queue"));
                e->queue_groups = groups;
            }
            HIPCHK(e, hipMemsetAsync(e->d_queue, 0, 4 * (size_t)(16 + groups), e->stream));
            c.q_ticket = e->d_queue; c.q_done = reinterpret_cast<int32_t*>(e->d_queue + 16); c.q_groups = groups;
            if ((int)grid.x > e->queue_capacity) grid.x = (unsigned)e->queue_capacity;
            queued = true;
        }
    }
    std::pair<hipEvent_t, hipEvent_t>* evp = nullptr;
    if (e->timing_on) { int rc_ev = env_next_events(e, &evp); if (rc_ev) return rc_ev; HIPCHK(e, hipEventRecord(evp->first, e->stream)); }
    if (queued) { DISPATCH_NC(e, k_env_collect_q, grid, block, e->d, c); }
    else DISPATCH_NC(e, k_env_collect, grid, block, e->d, c);
    if (evp) { HIPCHK(e, hipEventRecord(evp->second, e->stream)); e->ev0 = evp->first; e->ev1 = evp->second; e->timed = true; }
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}
int rlgpu_env_set_collect_queue(rlgpu_env* e, int mode) { if (mode < -1 || mode > 1) return RLGPU_ERR_ARG; e->queue_mode = mode; return RLGPU_OK; }
int rlgpu_collect(rlgpu_env* e, rlgpu_learner* l, int T, float* obs, int32_t* actions, float* logp, float* reward, int32_t* done, int deterministic) {
    return collect_impl(e, l, T, obs, actions, logp, reward, done, deterministic, 0, nullptr, "rlgpu_collect");
}
int rlgpu_collect_free(rlgpu_env* e, rlgpu_learner* l, int T_cap, int64_t target_agent_steps, float* obs, int32_t* actions, float* logp, float* reward, int32_t* done,
                       int32_t* steps_dev, int deterministic) {
    if (target_agent_steps <= 0) { e->err = "rlgpu_collect_free: target_agent_steps must be positive"; return RLGPU_ERR_ARG; }
    return collect_impl(e, l, T_cap, obs, actions, logp, reward, done, deterministic, target_agent_steps, steps_dev, "rlgpu_collect_free");
}

int rlgpu_env_overflow_counts(rlgpu_env* e, uint64_t* out5, int reset) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    unsigned int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef RLG_HAVE_OVERFLOW_COUNTS
    HIPCHK(e, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_overflow), sizeof(h)));
    if (reset) { const unsigned int z[5] = {0, 0, 0, 0, 0}; HIPCHK(e, hipMemcpyToSymbol(HIP_SYMBOL(g_overflow), z, sizeof(z))); }   // slots 0-4 only: 5 and 6 are rlgpu_env_epa_counts's
#else
    int d[64]; HIPCHK(e, hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d)));   // profiler build: the same events live in its scratch counters
    for (int i = 0; i < 5; i++) h[i] = (unsigned int)d[i];
    (void)reset;
#endif
    for (int i = 0; i < 5; i++) out5[i] = h[i];
    return RLGPU_OK;
}

// Contact points LOST since the last reset, process-wide: none, ever -- until round 4 a body touching a third mesh object with points and a car-car
// point beyond the env's pair pool lost theirs; since round 5 such a tick is redone with the big contact layout (tick_world_big), and the only
// thing that counts here is that layout itself running out of room, which the mesh format rules out (<= 32 objects).  Kept as an invariant to assert.
static int read_overflow_slot(rlgpu_env* e, int slot, uint64_t* out1, int reset) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    unsigned int h = 0;
#ifdef RLG_HAVE_OVERFLOW_COUNTS
    HIPCHK(e, hipMemcpyFromSymbol(&h, HIP_SYMBOL(g_overflow), sizeof(h), (size_t)slot * sizeof(unsigned int)));
    if (reset) { const unsigned int z = 0; HIPCHK(e, hipMemcpyToSymbol(HIP_SYMBOL(g_overflow), &z, sizeof(z), (size_t)slot * sizeof(unsigned int))); }
#else
    int d[64]; HIPCHK(e, hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d))); h = (unsigned int)d[slot]; (void)reset;
#endif
    *out1 = h;
    return RLGPU_OK;
}
int rlgpu_env_lost_contact_count(rlgpu_env* e, uint64_t* out1, int reset) { return read_overflow_slot(e, 7, out1, reset); }
// env-ticks whose contacts did not fit the LDS layout and were redone with the big one (same results as the reference, a few hundred microseconds each)
int rlgpu_env_big_layout_ticks(rlgpu_env* e, uint64_t* out1, int reset) { return read_overflow_slot(e, 10, out1, reset); }
int rlgpu_env_epa_counts(rlgpu_env* e, uint64_t* out2, int reset) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    unsigned int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#ifdef RLG_HAVE_OVERFLOW_COUNTS
    HIPCHK(e, hipMemcpyFromSymbol(h, HIP_SYMBOL(g_overflow), sizeof(h)));
    // reset: the counts as they stood are what is reported (as rlgpu_env_overflow_counts does); only the two EPA slots are cleared
    if (reset) { const unsigned int z2[2] = {0, 0}; HIPCHK(e, hipMemcpyToSymbol(HIP_SYMBOL(g_overflow), z2, sizeof(z2), 5 * sizeof(unsigned int))); }
#else
    int d[64]; HIPCHK(e, hipMemcpyFromSymbol(d, HIP_SYMBOL(g_dbg), sizeof(d)));
    h[5] = (unsigned int)d[5]; h[6] = (unsigned int)d[6]; (void)reset;
#endif
    out2[0] = h[5]; out2[1] = h[6];
    return RLGPU_OK;
}

#ifdef RLG_TICK_PROFILE
int rlgpu_env_debug_ints(rlgpu_env* e, int* out) {
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(int) * 64));
    return RLGPU_OK;
}
#ifdef RLG_FINE_PROF
int rlgpu_env_debug_fine(rlgpu_env* e, unsigned int* out, int n_blocks) {
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fine_blk), sizeof(unsigned int) * 32 * (size_t)std::min(n_blocks, 4096)));
    return RLGPU_OK;
}
#endif
// profiler build only: the 12 phase buckets (cycles) of the last k_env_step launch, 16 values per workgroup
int rlgpu_env_debug_step_prof(rlgpu_env* e, unsigned long long* out, int n_blocks) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_step_prof), sizeof(unsigned long long) * 16 * (size_t)std::min(n_blocks, 4096)));
    return RLGPU_OK;
}
#endif
// diagnostics (not part of rlgpu.h): per-workgroup {shader cycles, 100 MHz ticks} of `ticks` physics ticks; out has 10 * n_blocks entries (cycles, realtime, 8 phase accumulators of the PROFILE build)
int rlgpu_env_debug_tick_cycles(rlgpu_env* e, int ticks, unsigned long long* out, int cap_pairs, int* n_blocks) {
    HIPCHK(e, hipSetDevice(e->device));
    int nb = RLG_NC_PICK(e->nc, env_grid<2>(e->n_envs), env_grid<4>(e->n_envs), env_grid<6>(e->n_envs));
    *n_blocks = nb;
    if (nb > cap_pairs) return RLGPU_ERR_ARG;
    unsigned long long* dbuf = nullptr;
    HIPCHK(e, hipMalloc(&dbuf, sizeof(unsigned long long) * 10 * nb));
    dim3 grid(nb), block(WAVE * WPB);
    DISPATCH_NC(e, k_env_ticks, grid, block, e->d, ticks, dbuf);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, dbuf, sizeof(unsigned long long) * 10 * nb, hipMemcpyDeviceToHost));
    (void)hipFree(dbuf);
    return RLGPU_OK;
}

int rlgpu_env_set_controls(rlgpu_env* e, const float* controls_host) {
    if (!controls_host) return RLGPU_ERR_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bytes = sizeof(float) * 8 * (size_t)e->nc * (size_t)e->n_envs;
    float* dctl = nullptr;
    HIPCHK(e, hipMalloc(&dctl, bytes));
    HIPCHK(e, hipMemcpyAsync(dctl, controls_host, bytes, hipMemcpyHostToDevice, e->stream));
    dim3 grid((e->n_envs + MOVE_LANES - 1) / MOVE_LANES), block(64);
    DISPATCH_NC(e, k_set_controls, grid, block, e->d, dctl);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipFree(dctl));
    return RLGPU_OK;
}
int rlgpu_env_physics_ticks(rlgpu_env* e, int ticks) {
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid(RLG_NC_PICK(e->nc, env_grid<2>(e->n_envs), env_grid<4>(e->n_envs), env_grid<6>(e->n_envs))), block(WAVE * WPB);
    DISPATCH_NC(e, k_env_ticks, grid, block, e->d, ticks, (unsigned long long*)nullptr);
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_env_sync(rlgpu_env* e) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return RLGPU_OK;
}
int rlgpu_env_timing_total(rlgpu_env* e, float* total_ms, int* launches, int reset) {
    HIPCHK(e, hipSetDevice(e->device));
    if (e->ev_used) HIPCHK(e, hipEventSynchronize(e->ev_pool[e->ev_used - 1].second));   // the newest pair: whatever stream it was recorded on
    for (size_t i = 0; i < e->ev_used; i++) {
        float ms = 0.f; HIPCHK(e, hipEventElapsedTime(&ms, e->ev_pool[i].first, e->ev_pool[i].second));
        e->acc_ms += ms; e->acc_launches++;
    }
    e->ev_used = 0;
    if (total_ms) *total_ms = (float)e->acc_ms;
    if (launches) *launches = e->acc_launches;
    if (reset) { e->acc_ms = 0; e->acc_launches = 0; }
    return RLGPU_OK;
}
int rlgpu_env_last_step_ms(rlgpu_env* e, float* ms) {
    if (!e->timed) { *ms = 0.f; return RLGPU_ERR_STATE; }
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return RLGPU_OK;
}

}  // extern "C"
