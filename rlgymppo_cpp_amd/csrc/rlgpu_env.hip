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

#include "../../include/rlgpu.h"
#include "arena_gym.h"
#include "arena_mesh.h"

using namespace rlg;

static_assert(sizeof(RlgpuGymConfig) == sizeof(GymConfig), "C-ABI gym config must mirror rlg::GymConfig");

namespace {

constexpr int WAVE = 64;
constexpr int LDS_NODES = 2048;  // 64 KiB of BVH top levels per workgroup (160 KiB LDS per CU)

struct EnvDev {
    uint32_t* words;      // [n_words][n_envs]
    const BvhNode* nodes; const MeshTri* tris; int n_nodes, n_tris;
    const float* action_table;
    GymConfig cfg;
    int n_envs;
};

template <int NC>
__device__ void load_env(const EnvDev& d, int env, Arena<NC>& A, GymEnv<NC>& G) {
    WordReader r; r.base = d.words + env; r.stride = (size_t)d.n_envs; r.idx = 0;
    arena_visit(A, G, r);
    arena_finish_load(A);
}
template <int NC>
__device__ void store_env(const EnvDev& d, int env, Arena<NC>& A, GymEnv<NC>& G) {
    WordWriter w; w.base = d.words + env; w.stride = (size_t)d.n_envs; w.idx = 0;
    arena_visit(A, G, w);
}

__device__ MeshView stage_mesh(const EnvDev& d, BvhNode* lds_nodes) {
    int n_fast = d.n_nodes < LDS_NODES ? d.n_nodes : LDS_NODES;
    // 32-byte nodes copied as 2 x 16-byte vectors per lane: coalesced global reads, conflict-free ds_write_b128
    const float4* src = reinterpret_cast<const float4*>(d.nodes);
    float4* dst = reinterpret_cast<float4*>(lds_nodes);
    for (int i = threadIdx.x; i < n_fast * 2; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    MeshView mv; mv.nodes = d.nodes; mv.tris = d.tris; mv.nodes_fast = lds_nodes; mv.n_nodes = d.n_nodes; mv.n_tris = d.n_tris; mv.n_fast = n_fast;
    return mv;
}

template <int NC>
__global__ void __launch_bounds__(WAVE) k_env_step(EnvDev d, const int32_t* actions, float* next_obs, float* reward, int32_t* done) {
    __shared__ BvhNode lds_nodes[LDS_NODES];
    MeshView mv = stage_mesh(d, lds_nodes);
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    Arena<NC> A; GymEnv<NC> G;
    load_env(d, env, A, G);
    int32_t acts[NC]; float rew[NC]; int32_t dn;
    for (int k = 0; k < NC; k++) acts[k] = actions[(size_t)env * NC + k];
    const int D = obs_size<NC>();
    gym_step_env<NC>(A, G, d.cfg, mv, d.action_table, acts, (uint32_t)env, next_obs + (size_t)env * NC * D, (size_t)D, rew, &dn);
    for (int k = 0; k < NC; k++) { reward[(size_t)env * NC + k] = rew[k]; done[(size_t)env * NC + k] = dn; }
    store_env(d, env, A, G);
}

template <int NC>
__global__ void __launch_bounds__(WAVE) k_env_reset(EnvDev d, int run_setter, float* obs) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    Arena<NC> A; GymEnv<NC> G;
    load_env(d, env, A, G);
    const int D = obs_size<NC>();
    gym_reset_env<NC>(A, G, d.cfg, (uint32_t)env, obs ? obs + (size_t)env * NC * D : nullptr, (size_t)D, run_setter != 0);
    store_env(d, env, A, G);
}

template <int NC>
__global__ void __launch_bounds__(WAVE) k_env_ticks(EnvDev d, int ticks) {
    __shared__ BvhNode lds_nodes[LDS_NODES];
    MeshView mv = stage_mesh(d, lds_nodes);
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= d.n_envs) return;
    Arena<NC> A; GymEnv<NC> G;
    load_env(d, env, A, G);
    for (int t = 0; t < ticks; t++) { TickEvents ev; ev.bump_mask = 0; arena_tick(A, mv, d.cfg.seed_lo ^ 0xA511E9B3u, (uint32_t)env, ev); }
    store_env(d, env, A, G);
}

// AoS <-> SoA movers for the host fallback path
template <int NC>
__global__ void k_upload(EnvDev d, const RlgpuArenaState* src, const int32_t* env_ids, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int env = env_ids ? env_ids[i] : i;
    if (env < 0 || env >= d.n_envs) return;
    Arena<NC> A; GymEnv<NC> G;
    arena_from_host(A, G, src[i]);
    store_env(d, env, A, G);
}
template <int NC>
__global__ void k_download(EnvDev d, RlgpuArenaState* dst, const int32_t* env_ids, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int env = env_ids ? env_ids[i] : i;
    if (env < 0 || env >= d.n_envs) return;
    Arena<NC> A; GymEnv<NC> G;
    load_env(d, env, A, G);
    arena_to_host(A, G, dst[i]);
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
    EnvDev d{};
    BvhNode* d_nodes = nullptr; MeshTri* d_tris = nullptr; float* d_actions = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    std::string err;
};

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

int rlgpu_action_table(float* out, int cap_rows) {
    float tab[90 * 8];
    int n = build_action_table(tab);
    if (out) memcpy(out, tab, sizeof(float) * 8 * (size_t)std::min(n, cap_rows));
    return n;
}

int rlgpu_env_create(rlgpu_env** out, int device, int n_envs, int team_size, const RlgpuGymConfig* cfg) {
    if (!out || n_envs <= 0 || team_size < 1 || team_size > 3 || !cfg) return RLGPU_ERR_ARG;
    rlgpu_env* e = new rlgpu_env();
    *out = e;
    e->device = device; e->n_envs = n_envs; e->team_size = team_size; e->nc = 2 * team_size;
    HIPCHK(e, hipSetDevice(device));
    e->n_words = e->nc == 2 ? count_words<2>() : (e->nc == 4 ? count_words<4>() : count_words<6>());
    HIPCHK(e, hipMalloc(&e->d.words, e->n_words * (size_t)n_envs * 4));
    HIPCHK(e, hipMemset(e->d.words, 0, e->n_words * (size_t)n_envs * 4));
    float tab[90 * 8]; build_action_table(tab);
    HIPCHK(e, hipMalloc(&e->d_actions, sizeof(tab)));
    HIPCHK(e, hipMemcpy(e->d_actions, tab, sizeof(tab), hipMemcpyHostToDevice));
    e->d.action_table = e->d_actions;
    memcpy(&e->d.cfg, cfg, sizeof(GymConfig));
    e->d.n_envs = n_envs; e->d.nodes = nullptr; e->d.tris = nullptr; e->d.n_nodes = 0; e->d.n_tris = 0;
    HIPCHK(e, hipEventCreate(&e->ev0)); HIPCHK(e, hipEventCreate(&e->ev1));
    return RLGPU_OK;
}

void rlgpu_env_destroy(rlgpu_env* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->d.words) (void)hipFree(e->d.words);
    if (e->d_nodes) (void)hipFree(e->d_nodes);
    if (e->d_tris) (void)hipFree(e->d_tris);
    if (e->d_actions) (void)hipFree(e->d_actions);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
}
const char* rlgpu_env_last_error(const rlgpu_env* e) { return e ? e->err.c_str() : "null env"; }
int rlgpu_env_set_stream(rlgpu_env* e, void* s) { e->stream = (hipStream_t)s; return RLGPU_OK; }
int rlgpu_env_obs_size(const rlgpu_env* e) { return 51 + 19 * e->nc; }
int rlgpu_env_num_agents(const rlgpu_env* e) { return e->n_envs * e->nc; }
int rlgpu_env_num_actions(const rlgpu_env* e) { return e->d.cfg.n_actions; }
int rlgpu_env_state_words(const rlgpu_env* e) { return (int)e->n_words; }

int rlgpu_env_set_mesh(rlgpu_env* e, const float* verts, int n_verts, const int32_t* tris, int n_tris) {
    HIPCHK(e, hipSetDevice(e->device));
    HostMesh m = build_host_mesh(verts, n_verts, tris, n_tris);
    if (e->d_nodes) { (void)hipFree(e->d_nodes); e->d_nodes = nullptr; }
    if (e->d_tris) { (void)hipFree(e->d_tris); e->d_tris = nullptr; }
    e->d.n_nodes = (int)m.nodes.size(); e->d.n_tris = (int)m.tris.size();
    if (!m.nodes.empty()) {
        HIPCHK(e, hipMalloc(&e->d_nodes, m.nodes.size() * sizeof(BvhNode)));
        HIPCHK(e, hipMalloc(&e->d_tris, m.tris.size() * sizeof(MeshTri)));
        HIPCHK(e, hipMemcpy(e->d_nodes, m.nodes.data(), m.nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
        HIPCHK(e, hipMemcpy(e->d_tris, m.tris.data(), m.tris.size() * sizeof(MeshTri), hipMemcpyHostToDevice));
    }
    e->d.nodes = e->d_nodes; e->d.tris = e->d_tris;
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
    for (auto& f : files) {
        std::ifstream in(f, std::ios::binary);
        std::vector<uint8_t> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        if (!append_cmf(buf.data(), buf.size(), v, t)) { e->err = "bad cmf file " + f; return RLGPU_ERR_ARG; }
    }
    return rlgpu_env_set_mesh(e, v.data(), (int)v.size() / 3, t.data(), (int)t.size() / 3);
}

#define DISPATCH_NC(e, KERNEL, grid, block, ...)                                                             \
    do {                                                                                                     \
        if ((e)->nc == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, (e)->stream, __VA_ARGS__);          \
        else if ((e)->nc == 4) hipLaunchKernelGGL((KERNEL<4>), grid, block, 0, (e)->stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL((KERNEL<6>), grid, block, 0, (e)->stream, __VA_ARGS__);                       \
    } while (0)

int rlgpu_env_upload_states(rlgpu_env* e, const RlgpuArenaState* host, const int32_t* env_ids, int n) {
    if (n <= 0) return RLGPU_OK;
    HIPCHK(e, hipSetDevice(e->device));
    RlgpuArenaState* dsrc = nullptr; int32_t* dids = nullptr;
    HIPCHK(e, hipMalloc(&dsrc, sizeof(RlgpuArenaState) * (size_t)n));
    HIPCHK(e, hipMemcpyAsync(dsrc, host, sizeof(RlgpuArenaState) * (size_t)n, hipMemcpyHostToDevice, e->stream));
    if (env_ids) { HIPCHK(e, hipMalloc(&dids, 4 * (size_t)n)); HIPCHK(e, hipMemcpyAsync(dids, env_ids, 4 * (size_t)n, hipMemcpyHostToDevice, e->stream)); }
    dim3 grid((n + 63) / 64), block(64);
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
    HIPCHK(e, hipMemsetAsync(ddst, 0, sizeof(RlgpuArenaState) * (size_t)n, e->stream));
    if (env_ids) { HIPCHK(e, hipMalloc(&dids, 4 * (size_t)n)); HIPCHK(e, hipMemcpyAsync(dids, env_ids, 4 * (size_t)n, hipMemcpyHostToDevice, e->stream)); }
    dim3 grid((n + 63) / 64), block(64);
    DISPATCH_NC(e, k_download, grid, block, e->d, ddst, (const int32_t*)dids, n);
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipMemcpyAsync(host, ddst, sizeof(RlgpuArenaState) * (size_t)n, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    (void)hipFree(ddst); if (dids) (void)hipFree(dids);
    return RLGPU_OK;
}

int rlgpu_env_reset(rlgpu_env* e, int run_setter, float* obs_dev) {
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid((e->n_envs + WAVE - 1) / WAVE), block(WAVE);
    DISPATCH_NC(e, k_env_reset, grid, block, e->d, run_setter, obs_dev);
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_env_step(rlgpu_env* e, const int32_t* actions, float* next_obs, float* reward, int32_t* done) {
    if (!actions || !next_obs || !reward || !done) { e->err = "rlgpu_env_step: null device pointer"; return RLGPU_ERR_ARG; }
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid((e->n_envs + WAVE - 1) / WAVE), block(WAVE);
    HIPCHK(e, hipEventRecord(e->ev0, e->stream));
    DISPATCH_NC(e, k_env_step, grid, block, e->d, actions, next_obs, reward, done);
    HIPCHK(e, hipEventRecord(e->ev1, e->stream));
    e->timed = true;
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_env_physics_ticks(rlgpu_env* e, int ticks) {
    HIPCHK(e, hipSetDevice(e->device));
    dim3 grid((e->n_envs + WAVE - 1) / WAVE), block(WAVE);
    DISPATCH_NC(e, k_env_ticks, grid, block, e->d, ticks);
    HIPCHK(e, hipGetLastError());
    return RLGPU_OK;
}

int rlgpu_env_sync(rlgpu_env* e) {
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return RLGPU_OK;
}
int rlgpu_env_last_step_ms(rlgpu_env* e, float* ms) {
    if (!e->timed) { *ms = 0.f; return RLGPU_ERR_STATE; }
    HIPCHK(e, hipEventSynchronize(e->ev1));
    HIPCHK(e, hipEventElapsedTime(ms, e->ev0, e->ev1));
    return RLGPU_OK;
}

}  // extern "C"
