// Learner.hip -- C++ host side of the reference's API (include/RLGymPPO_CPP/Learner.h) over the C-ABI of include/rlgpu.h.
// What it restates (reference file:line, PUB = RLGymPPO_CPP/src/public/RLGymPPO_CPP, PRIV = .../src/private/RLGymPPO_CPP):
//   Learner::Learner            PUB/Learner.cpp:17-156       one device env batch + device learner instead of agent threads + libtorch
//   Learner::Learn              PUB/Learner.cpp:436-606      collect -> AddNewExperience -> PPOLearner::Learn -> report -> callbacks -> save
//   Learner::AddNewExperience   PUB/Learner.cpp:608-703
//   PPOLearner::Learn           PRIV/PPO/PPOLearner.cpp:67-349
//   Learner::Save / Load        PUB/Learner.cpp:171-376, PRIV/PPO/PPOLearner.cpp:362-502   (same folder layout and file names;
//                                .lt payloads = the reference's TorchScript zip archives, through rlgpu_lt_* of librlgpu.so)
// Everything on the data path stays in device memory; this file only sequences launches.  Compiled by hipcc because of the
// three small bookkeeping kernels below.
#include <unistd.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cstring>
#include <fstream>
#include <future>
#include <numeric>
#include <thread>

#include <RLGymPPO_CPP/Learner.h>
#include "../../include/rlgpu_state.h"
#include "host_util.h"
#include "HostEnvPath.h"

namespace {

// done (int32) -> float, and the collector's truncation mark: the last step of every trajectory is truncated unless done
// (ThreadAgentManager.cpp:55)
__global__ void k_done_trunc(const int32_t* done, int T, int n, float* done_f, float* trunc) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)T * n) return;
    float d = done[i] ? 1.f : 0.f;
    done_f[i] = d;
    trunc[i] = (i / n == (size_t)(T - 1)) ? 1.f - d : 0.f;
}
// out[k] += sum |x_k| for up to three arrays (report averages, Learner.cpp:670-677)
__global__ void k_abs_sums(const float* a, const float* b, const float* c, size_t n, float* out) {
    float s[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { s[0] += fabsf(a[i]); s[1] += fabsf(b[i]); s[2] += fabsf(c[i]); }
    for (int k = 0; k < 3; k++) {
        float v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&out[k], v);
    }
}
__global__ void k_sum(const float* a, size_t n, float* out) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}

// ---- iterations whose trajectories have their own lengths (free-running collection) ----
__global__ void k_fill_i32(int32_t* p, int n, int32_t v) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
__global__ void k_done_f(const int32_t* done, size_t n, float* done_f) { size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (i < n) done_f[i] = done[i] ? 1.f : 0.f; }
// every player's current observation = the row after ITS OWN last step becomes row block 0 of the next launch
__global__ void k_carry_obs(float* obs, const int32_t* steps, int players, int nAgents, int D) {
    const int a = blockIdx.x;
    const size_t src = ((size_t)steps[a / players] * nAgents + a) * D, dst = (size_t)a * D;
    if (src != dst) for (int i = threadIdx.x; i < D; i += blockDim.x) obs[dst + i] = obs[src + i];
}
// out[0..2] += sum |ret|, |adv|, |tgt|, out[3] += sum rew over the rows that exist: t < steps[a / players]
__global__ void k_ragged_sums(const float* ret, const float* adv, const float* tgt, const float* rew, const int32_t* steps, int players, int Tused, int n, float* out) {
    float s[4] = {0, 0, 0, 0};
    const size_t total = (size_t)Tused * n;
    const size_t nt = (size_t)gridDim.x * blockDim.x, g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nt % (size_t)n == 0) {
        // a thread keeps its column (one division per thread, not per element: the 64-bit i / n, i % n of the general loop below were most of
        // this kernel's 57 us) and walks down the steps
        const int a = (int)(g % (size_t)n), dt = (int)(nt / (size_t)n), lim = min(Tused, steps[a / players]);
        for (int t = (int)(g / (size_t)n); t < lim; t += dt) { const size_t i = (size_t)t * n + a; s[0] += fabsf(ret[i]); s[1] += fabsf(adv[i]); s[2] += fabsf(tgt[i]); s[3] += rew[i]; }
    } else {
        for (size_t i = g; i < total; i += nt) {
            const int t = (int)(i / n), a = (int)(i % n);
            if (t < steps[a / players]) { s[0] += fabsf(ret[i]); s[1] += fabsf(adv[i]); s[2] += fabsf(tgt[i]); s[3] += rew[i]; }
        }
    }
    // one atomic per workgroup and value (four same-address atomics per WAVEFRONT were most of this kernel's 57 us)
    __shared__ float part[4][4];
    for (int k = 0; k < 4; k++) {
        float v = s[k];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 4) { float v = 0.f; for (int w = 0; w < (int)(blockDim.x >> 6); w++) v += part[w][threadIdx.x]; atomicAdd(&out[threadIdx.x], v); }
}

std::filesystem::path g_mesh_folder;

// Debug mode RLGPU_REDZONE=<bytes> (as in librlgpu.so: rlgpu_env_check_redzones): guard bytes behind every device buffer this library owns;
// ~Learner checks them, the env batch's and the learner's, and says so on stderr -- or ends the process with the name of what was overwritten.
struct HostRedzone { void* base; size_t bytes; };
std::vector<HostRedzone> g_redzones;
size_t redzone_bytes() { const char* s = std::getenv("RLGPU_REDZONE"); return s ? (size_t)atol(s) : 0; }
template <class T>
T* dev_alloc(size_t n) {
    T* p = nullptr; const size_t bytes = std::max<size_t>(n, 1) * sizeof(T), rz = redzone_bytes();
    HOST_HIP(hipMalloc(&p, bytes + rz));
    if (rz) { HOST_HIP(hipMemset((char*)p + bytes, 0xC5, rz)); g_redzones.push_back({(void*)p, bytes}); }
    return p;
}
void dev_release(void* p) {
    for (size_t i = 0; i < g_redzones.size(); i++) if (g_redzones[i].base == p) { g_redzones.erase(g_redzones.begin() + (long)i); break; }
    (void)hipFree(p);
}
// returns "" or what was overwritten
std::string host_redzones_check() {
    const size_t rz = redzone_bytes();
    std::vector<unsigned char> h(rz);
    for (size_t k = 0; k < g_redzones.size(); k++) {
        HOST_HIP(hipMemcpy(h.data(), (const char*)g_redzones[k].base + g_redzones[k].bytes, rz, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < rz; i++) if (h[i] != 0xC5)
            return "host-library device buffer #" + std::to_string(k) + " (" + std::to_string(g_redzones[k].bytes) + " bytes, allocation order of Learner.hip) overwritten at +" + std::to_string(i) + " past its end";
    }
    return "";
}

}  // namespace

namespace RocketSim {
void Init(const std::filesystem::path& collisionMeshesFolder, bool silent) {
    g_mesh_folder = collisionMeshesFolder;
    if (!silent) RG_LOG("RocketSim::Init: arena meshes from \"" << (collisionMeshesFolder / "soccar").string() << "\" (procedural soccar mesh if absent)");
}
const std::filesystem::path& GetCollisionMeshFolder() { return g_mesh_folder; }
}

namespace RLGSC {
void LoadArenaMesh(rlgpu_env* env, bool quiet) {
    auto check = [&](int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("rlgpu_env_" << what << " failed (" << rc << "): " << rlgpu_env_last_error(env)); };
    const std::filesystem::path soccar = RocketSim::GetCollisionMeshFolder() / "soccar";
    if (!RocketSim::GetCollisionMeshFolder().empty() && std::filesystem::is_directory(soccar)) check(rlgpu_env_load_cmf_dir(env, soccar.string().c_str()), "load_cmf_dir");
    else {
        if (!quiet) RG_LOG("Learner: no collision meshes at \"" << soccar.string() << "\" -- using the procedural soccar mesh");
        check(rlgpu_env_set_procedural_mesh(env), "set_procedural_mesh");
    }
}
}  // namespace RLGSC

namespace RLGPC {

struct Learner::Impl {
    rlgpu_env* env = nullptr; rlgpu_learner* lrn = nullptr; rlgpu_shuffler* shuf = nullptr;
    rlgpu_comm* comm = nullptr; int rank = 0, world = 1, device = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> arEvents; size_t arUsed = 0; double arMs = 0; int arCalls = 0; bool arTimed = false;   // all-reduce clocks (bench)
    float* retShare = nullptr;   // rank 0's first returns, broadcast so every rank feeds the same statistic
    RLGSC::Match* match = nullptr; RLGSC::Gym* gym = nullptr;
    int nEnvs = 0, nAgents = 0, nPlayers = 0, D = 0, A = 0, T = 0, tickSkip = 8;
    int64_t B = 0, batch = 0, mini = 0; int maxRows = 0;
    float *obs = nullptr, *logp = nullptr, *rew = nullptr, *doneF = nullptr, *trunc = nullptr, *vals = nullptr, *adv = nullptr, *tgt = nullptr, *ret = nullptr,
          *metrics = nullptr, *scratch = nullptr;
    int32_t *acts = nullptr, *done = nullptr, *idx = nullptr;
    // ExperienceBuffer (ExperienceBuffer.h): the FIFO's iterations stay in device slots of B rows, the library tracks the live rows
    rlgpu_expbuf* fifo = nullptr;
    float *exObs = nullptr, *exLogp = nullptr, *exAdv = nullptr, *exTgt = nullptr; int32_t* exActs = nullptr;
    bool first = true, renderOnly = false, fusedCollect = true;
    // free-running collection (LearnerConfig::lockstepCollection = false; rlgpu_collect_free): buffers and FIFO slots are laid out for Tcap steps,
    // steps[e] = what env e made in the last launch (== T after a lockstep launch), trajOff[slot] = where each trajectory of the iteration in
    // that FIFO slot starts in the concatenated batch
    bool ragged = false, freeOk = true, lastFree = false; int Tcap = 0, Tused = 0; int64_t lastRows = 0, lastRowsAll = 0;   // lastRowsAll: the iteration's timesteps of ALL ranks
    int32_t *steps = nullptr, *trajOff = nullptr, *permDev = nullptr;
    std::vector<int32_t> hSteps, hAgentSteps, perm[2]; int permFlip = 0;
    std::future<int64_t> permDraw; bool permPending = false; std::string permEngine;   // a permutation drawn ahead for a predicted FIFO size + the engine before it
    void StartPermDraw(int64_t size) {
        char buf[256];
        if (rlgpu_shuffler_get_state(shuf, buf, sizeof(buf)) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: shuffler state");
        permEngine = buf;
        int32_t* out = perm[permFlip].data();
        permDraw = std::async(std::launch::async, [this, out, size]() -> int64_t { return rlgpu_shuffler_next_i32(shuf, size, out) == RLGPU_OK ? size : -1; });
        permPending = true;
    }
    // the permutation of [0, size): the one drawn ahead when it was drawn for this size, else the engine goes back and draws again
    const int32_t* TakePerm(int64_t size) {
        if (permPending) {
            permPending = false;
            const int64_t got = permDraw.get();
            if (got < 0) RG_ERR_CLOSE("ExperienceBuffer: shuffle failed");
            if (got == size) { const int32_t* r = perm[permFlip].data(); permFlip ^= 1; return r; }
            if (rlgpu_shuffler_set_state(shuf, permEngine.c_str()) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: shuffler state");
        }
        if (rlgpu_shuffler_next_i32(shuf, size, perm[permFlip].data()) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: shuffle failed");
        const int32_t* r = perm[permFlip].data(); permFlip ^= 1; return r;
    }
    Timer renderTimer;
    uint64_t cumulativeModelUpdates = 0, tsSinceSave = 0;
    // collectionDuringLearn (LearnerConfig.h:46-50, Learner.cpp:473-510): the PPO epochs of iteration k run on their own stream while iteration
    // k + 1 is collected; their statistics reach the report one iteration later
    hipStream_t learnStream = nullptr; hipEvent_t evReady = nullptr, evLearnDone = nullptr;
    bool learnPending = false; int pendMini = 0; Timer pendTimer;
    uint32_t envStreamEpoch = 0;   // second key word of the env batch's RNG streams: bumped on every resume (no replay of the resets of the run continued)
    std::vector<GameInst> games;
    // ---- host path: plugin kinds without a device form (Match::DevicePlan) and step callbacks: HostEnvPath.h (shared with the skill tracker) ----
    RLGSC::Match::DevicePlan plan;
    int Ddev = 0;
    HostEnvPath hp;
    void SetupHostPath(const EnvCreateFn& create, int numThreads);
    void HostStep(Learner* self, int t);
    // LearnerConfig::deferHostRewards: a user reward function is the only host plugin (deferredReward; the step kernel then resets ended envs itself), or
    // there is none and a step callback is installed -- collection stays fused and the host work is replayed after the launch (HostEnvPath::ReplayCollected)
    bool deferredReward = false;
    bool DeferredNow(const Learner* self) const;
    void DeferredReplay(Learner* self);
    // the permutation of the NEXT epoch is drawn by a worker while this thread launches the current one (a draw depends on the
    // FIFO's bookkeeping only): 2-3 ms of std::shuffle per 262 144 rows that would otherwise leave the GPU idle
    std::vector<int32_t> phys[2]; int physFlip = 0;
    std::future<int64_t> nextDraw; bool drawPending = false;
    int pendingSlot = -1;
    void StartDraw() {
        int32_t* buf = phys[physFlip].data();
        nextDraw = std::async(std::launch::async, [this, buf]() -> int64_t {
            const int64_t cur = rlgpu_expbuf_size(fifo);
            return rlgpu_expbuf_shuffled_rows(fifo, shuf, buf) == RLGPU_OK ? cur : -1;
        });
        drawPending = true;
    }
    int64_t TakeDraw() {
        if (!drawPending) StartDraw();
        drawPending = false;
        const int64_t cur = nextDraw.get();
        if (cur < 0) RG_ERR_CLOSE("ExperienceBuffer: shuffle failed");
        return cur;
    }
    Timer iterTimer;

    void EnvCheck(int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("rlgpu_env_" << what << " failed (" << rc << "): " << rlgpu_env_last_error(env)); }
    void LrnCheck(int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("rlgpu_" << what << " failed (" << rc << "): " << rlgpu_learner_last_error(lrn)); }
    float* ObsAt(int t) { return obs + (size_t)t * nAgents * D; }
};

Learner::Learner(EnvCreateFn envCreateFn_, LearnerConfig config_) : config(config_), envCreateFn(envCreateFn_), impl(new Impl()) {
    Impl& m = *impl;
    if (config.deviceType == LearnerDeviceType::CPU) RG_ERR_CLOSE("LearnerDeviceType::CPU: this build has no CPU path (the hot path is HIP kernels)");
    if (config.standardizeOBS) RG_ERR_CLOSE("LearnerConfig.standardizeOBS has not yet been implemented, sorry");   // Learner.cpp:33-34
    if (config.timestepsPerSave == 0) config.timestepsPerSave = config.timestepsPerIteration;
    m.renderOnly = config.renderMode && !config.renderDuringTraining;
    if (m.renderOnly) {   // Learner.cpp:38-52: one game, no metrics, no checkpoints, never learns
        RG_LOG("\tRender mode is enabled, overriding:");
        config.numThreads = config.numGamesPerThread = 1; RG_LOG("\t > numThreads, numGamesPerThread = 1");
        config.sendMetrics = false; RG_LOG("\t > sendMetrics = false");
        config.checkpointSaveFolder.clear(); RG_LOG("\t > checkpointSaveFolder = none");
        config.timestepsPerIteration = 1; RG_LOG("\t > timestepsPerIteration = inf (the render loop never hands a batch to the learner)");
    }
    // The reference calls envCreateFn once per game and once more to probe the obs size (Learner.cpp:99-109); every call
    // describes the same env, so one call is enough to configure the whole device batch.
    {   // multi-GPU launch? (torchrun or any launcher exporting RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT)
        const char* ws = std::getenv("WORLD_SIZE");
        if (ws && std::atoi(ws) > 1) {
            if (rlgpu_comm_init_env(&m.comm, &m.rank, &m.world) != RLGPU_OK) RG_ERR_CLOSE("rlgpu_comm_init_env failed: " << rlgpu_comm_last_error(nullptr));
            m.device = rlgpu_comm_device(m.comm);   // LOCAL_RANK's device: the one the communicator is bound to (one source for both)
        }
        HOST_HIP(hipSetDevice(m.device));
    }
    EnvCreateResult ecr = envCreateFn();
    if (!ecr.match || !ecr.gym) RG_ERR_CLOSE("EnvCreateFn returned a null match or gym");
    m.match = ecr.match; m.gym = ecr.gym; m.tickSkip = ecr.gym->tickSkip;
    m.plan = m.match->PlanDevice(m.tickSkip);   // which plugin kinds the step kernel runs, which stay on the host
    RlgpuGymConfig gcfg = m.plan.cfg;
    m.deferredReward = config.deferHostRewards && m.plan.hostReward && !m.plan.hostTerminal && !m.plan.hostObs && !m.plan.hostSetter && !m.plan.hostParser && m.match->spawnOpponents;
    if (m.deferredReward) { gcfg.host_resets = 0; m.plan.cfg.host_resets = 0; m.hp.deviceResets = true; }   // the terminal conditions and the setter are the device's: an ended env is reset inside the step
    gcfg.seed_lo = (uint32_t)config.randomSeed + 1000u * (uint32_t)m.rank; gcfg.seed_hi = 0;   // every rank its own env RNG streams; rank 0 = the single-GPU run
    m.nEnvs = config.numThreads * config.numGamesPerThread;
    m.nPlayers = m.match->playerAmount;
    int rc = rlgpu_env_create(&m.env, m.device, m.nEnvs, m.match->teamSize, &gcfg);
    m.EnvCheck(rc, "create");
    if (!ecr.gym->arena->GetMutatorConfig().IsDefault()) {   // Gym(match, tickSkip, carConfig, gameMode, mutatorConfig): every game of the batch runs under the probe env's mutators (Gym.cpp:40-44)
        const RlgpuMutators mut = ecr.gym->arena->GetMutatorConfig().ToDevice();
        m.EnvCheck(rlgpu_env_set_mutators(m.env, &mut), "set_mutators");
    }
    RLGSC::LoadArenaMesh(m.env, m.rank != 0 || std::getenv("RLGPU_QUIET"));
    if (config.deviceStepMetrics) m.EnvCheck(rlgpu_env_enable_step_stats(m.env, 1), "enable_step_stats");
    m.nAgents = rlgpu_env_num_agents(m.env); m.Ddev = m.D = rlgpu_env_obs_size(m.env); m.A = rlgpu_env_num_actions(m.env);
    if (m.plan.AnyHost()) {
        if (m.rank == 0 && !std::getenv("RLGPU_QUIET"))
            RG_LOG("Learner: plugins without a device form run on the host " << (m.deferredReward ? "after every collection launch (" : "every step (") << (m.plan.hostReward ? "reward " : "") << (m.plan.hostTerminal ? "terminal-conditions " : "")
                   << (m.plan.hostObs ? "obs-builder " : "") << (m.plan.hostSetter ? "state-setter " : "") << (m.plan.hostParser ? "action-parser " : "") << "); the rest stays on the device");
        m.SetupHostPath(envCreateFn, config.numThreads);
        if (m.plan.hostObs) {   // the obs width is whatever the user's builder returns (Learner.cpp:99-109 probes it the same way)
            m.EnvCheck(rlgpu_env_reset(m.env, 1, m.hp.devObs), "reset");
            RlgpuArenaState st; const int32_t env0 = 0;
            m.EnvCheck(rlgpu_env_download_states(m.env, &st, &env0, 1), "download_states");
            RLGSC::GameState gs(st, (int)st.tick_count);
            m.match->EpisodeReset(gs);
            m.D = (int)m.match->BuildObservations(gs).at(0).size();
            m.hp.D = m.D;
        }
    }
    obsSize = m.D; actionAmount = m.A;
    // steps per env and iteration: enough whole steps of the batch to reach timestepsPerIteration (ThreadAgent.cpp:158-163 stops
    // each thread once it passed its share)
    m.T = (int)std::max<int64_t>(1, (config.timestepsPerIteration + m.nAgents - 1) / m.nAgents);
    m.B = (int64_t)m.T * m.nAgents;
    m.batch = config.ppo.batchSize > 0 ? std::min<int64_t>(config.ppo.batchSize, m.B) : m.B;
    m.mini = config.ppo.miniBatchSize > 0 ? std::min<int64_t>(config.ppo.miniBatchSize, m.batch) : m.batch;
    if (m.batch % m.mini != 0) RG_ERR_CLOSE("PPOLearner: batchSize (" << m.batch << ") must be a multiple of miniBatchSize (" << m.mini << ")");   // PPOLearner.cpp:31-33
    m.maxRows = (int)std::max<int64_t>(m.mini, m.nAgents);

    RlgpuLearnerConfig lc{};
    lc.obs_size = m.D; lc.n_actions = m.A;
    if (config.ppo.policyLayerSizes.size() > 8 || config.ppo.criticLayerSizes.size() > 8) RG_ERR_CLOSE("at most 8 hidden layers per network");
    lc.n_policy_layers = (int)config.ppo.policyLayerSizes.size(); lc.n_critic_layers = (int)config.ppo.criticLayerSizes.size();
    for (int i = 0; i < lc.n_policy_layers; i++) lc.policy_layers[i] = config.ppo.policyLayerSizes[i];
    for (int i = 0; i < lc.n_critic_layers; i++) lc.critic_layers[i] = config.ppo.criticLayerSizes[i];
    lc.policy_lr = config.ppo.policyLR; lc.critic_lr = config.ppo.criticLR; lc.ent_coef = config.ppo.entCoef; lc.clip_range = config.ppo.clipRange;
    lc.temperature = config.ppo.policyTemperature; lc.use_bf16 = config.ppo.autocastLearn ? (std::getenv("RLGPU_AUTOCAST_FP16") ? 2 : 1) : 0;   // autocastLearn: bf16 operands (the reference's autocast dtype, FrameworkTorch.h:14); RLGPU_AUTOCAST_FP16=1: fp16 operands + dynamic loss scale in the minibatch kernels (BASELINE configs[4]'s wording)
    lc.seed_lo = (uint32_t)config.randomSeed; lc.seed_hi = 0; lc.max_rows = m.maxRows;
    rc = rlgpu_learner_create(&m.lrn, m.device, &lc);
    m.LrnCheck(rc, "learner_create");
    // identical parameters on every rank (same init seed), independent exploration: the action sampler is keyed on the rank
    m.LrnCheck(rlgpu_learner_set_sampler(m.lrn, (uint32_t)m.rank, 0), "learner_set_sampler");
    m.retShare = dev_alloc<float>((size_t)std::max(1, config.maxReturnsPerStatsInc));
    rlgpu_shuffler_create(&m.shuf, (uint32_t)config.randomSeed);

    // free-running collection is tried when nothing needs the host between steps and the games have both teams (the fused launch's row mapping);
    // whether the batch is resident at once is known at the first launch (rlgpu_collect_free: RLGPU_ERR_STATE -> lockstep from then on)
    if (std::getenv("RLGPU_LOCKSTEP_COLLECTION")) config.lockstepCollection = true;   // (tests that count timesteps exactly)
    m.EnvCheck(rlgpu_env_set_collect_queue(m.env, config.collectStepQueue), "set_collect_queue");
    if (config.deterministicGradients) { config.lockstepCollection = true; m.LrnCheck(rlgpu_learner_set_deterministic(m.lrn, 1), "learner_set_deterministic"); }
    m.ragged = !config.lockstepCollection && !m.renderOnly && (!m.plan.AnyHost() || m.deferredReward) && !config.renderMode && m.match->spawnOpponents && !config.deterministic;
    m.Tcap = m.ragged ? 2 * m.T : m.T;
    const size_t TN = (size_t)m.Tcap * m.nAgents;
    m.obs = dev_alloc<float>((size_t)(m.Tcap + 1) * m.nAgents * m.D);
    m.acts = dev_alloc<int32_t>(TN); m.done = dev_alloc<int32_t>(TN);
    m.steps = dev_alloc<int32_t>((size_t)m.nEnvs); m.hSteps.assign((size_t)m.nEnvs, m.T); m.hAgentSteps.assign((size_t)m.nAgents, m.T);
    hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((m.nEnvs + 255) / 256)), dim3(256), 0, nullptr, m.steps, m.nEnvs, (int32_t)m.T);
    m.Tused = m.T; m.lastRows = m.B; m.lastRowsAll = (int64_t)m.B * m.world;
    if (m.ragged) {
        if (rlgpu_expbuf_create_ragged(&m.fifo, config.expBufferSize, m.Tcap, m.nAgents, m.B) != RLGPU_OK) {
            RG_LOG("\tfree-running collection: expBufferSize " << config.expBufferSize << " would keep more than 15 iterations resident -> lockstep collection");
            m.ragged = false; m.Tcap = m.T;
        }
    }
    if (!m.ragged && rlgpu_expbuf_create(&m.fifo, config.expBufferSize, m.T, m.nAgents) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: bad expBufferSize " << config.expBufferSize);
    ppo = new PPOLearner(); ppo->device = m.lrn;
    agentMgr = new ThreadAgentManager(); agentMgr->device = m.env; agentMgr->numGames = m.nEnvs;
    expBuffer = new ExperienceBuffer(); expBuffer->device = m.fifo;
    const size_t EX = (size_t)rlgpu_expbuf_num_slots(m.fifo) * TN;
    m.idx = dev_alloc<int32_t>(EX);
    m.exObs = dev_alloc<float>(EX * m.D); m.exActs = dev_alloc<int32_t>(EX); m.exLogp = dev_alloc<float>(EX); m.exAdv = dev_alloc<float>(EX); m.exTgt = dev_alloc<float>(EX);
    m.logp = dev_alloc<float>(TN); m.rew = dev_alloc<float>(TN); m.doneF = dev_alloc<float>(TN); m.trunc = dev_alloc<float>(TN);
    m.adv = dev_alloc<float>(TN); m.tgt = dev_alloc<float>(TN); m.ret = dev_alloc<float>(TN);
    m.vals = dev_alloc<float>(TN + m.nAgents); m.metrics = dev_alloc<float>(8); m.scratch = dev_alloc<float>(8);
    m.phys[0].resize(EX); m.phys[1].resize(EX);
    if (m.ragged) {
        m.trajOff = dev_alloc<int32_t>((size_t)rlgpu_expbuf_num_slots(m.fifo) * (m.nAgents + 1)); m.permDev = dev_alloc<int32_t>(EX);
        m.perm[0].resize(EX); m.perm[1].resize(EX);
    }
    if (!m.plan.hostObs) m.EnvCheck(rlgpu_env_reset(m.env, 1, m.ObsAt(0)), "reset");
    if (m.plan.AnyHost()) {
        std::vector<int32_t> all(m.nEnvs); std::iota(all.begin(), all.end(), 0);
        m.hp.ResetEnvs(all, m.ObsAt(0), true);
    }

    if (config.saveFolderAddUnixTimestamp && !config.checkpointSaveFolder.empty())
        config.checkpointSaveFolder += "-" + std::to_string(std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count());
    if (config.renderMode) renderSender = new RenderSender();                                                                       // Learner.cpp:128-135
    if (config.skillTrackerConfig.enabled) {                                                                                          // Learner.cpp:136-143
        if (config.skillTrackerConfig.envCreateFunc == NULL) config.skillTrackerConfig.envCreateFunc = envCreateFn;
        skillTracker = new SkillTracker(config.skillTrackerConfig, m.lrn, m.D, m.A, config.ppo.policyLayerSizes, config.randomSeed, renderSender);
    }
    if (config.collectionDuringLearn && !m.renderOnly && !rlgpu_learner_inference_is_standalone(m.lrn)) {
        // inference would share the learner's activation scratch with the PPO minibatches running on the other stream (fp32 mode, nets too
        // wide for the fused inference kernel): collection and learning stay in sequence, as with collectionDuringLearn = false
        RG_LOG("\tcollectionDuringLearn: inference is not standalone in this configuration (autocastLearn = false or wide nets) -> collection pauses during learning");
    } else if (config.collectionDuringLearn && !m.renderOnly) {
        HOST_HIP(hipStreamCreateWithFlags(&m.learnStream, hipStreamNonBlocking));
        HOST_HIP(hipEventCreateWithFlags(&m.evReady, hipEventDisableTiming)); HOST_HIP(hipEventCreateWithFlags(&m.evLearnDone, hipEventDisableTiming));
    }
    if (!config.checkpointLoadFolder.empty()) Load();
    // replicas: whatever the seeds and the checkpoint files made of them, every rank goes on from rank 0's parameters, Adam moments and step counters
    if (m.comm) m.LrnCheck(rlgpu_learner_sync_from_rank0(m.lrn, m.comm), "learner_sync_from_rank0");
    if (config.sendMetrics && m.rank == 0) {                                                                                          // Learner.cpp:149-155
        if (!runID.empty()) RG_LOG("\tRun ID: " << runID);
        metricSender = new MetricSender(config.metricsProjectName, config.metricsGroupName, config.metricsRunName, runID);
    }
}

Learner::~Learner() {
    Impl& m = *impl;
    if (redzone_bytes()) {      // debug mode: did anything write past a device buffer during this run?
        (void)hipDeviceSynchronize();
        std::string bad = host_redzones_check();
        if (bad.empty() && m.env && rlgpu_env_check_redzones(m.env) != RLGPU_OK) bad = std::string("env batch: ") + rlgpu_env_last_error(m.env);
        if (bad.empty() && m.lrn && rlgpu_learner_check_redzones(m.lrn) != RLGPU_OK) bad = std::string("learner: ") + rlgpu_learner_last_error(m.lrn);
        if (!bad.empty()) { fprintf(stderr, "RLGPU_REDZONE: %s\n", bad.c_str()); fflush(stderr); _exit(3); }
        fprintf(stderr, "RLGPU_REDZONE: clean (%zu host-library buffers, the env batch, the learner)\n", g_redzones.size());
        g_redzones.clear();
    }
    for (void* p : {(void*)m.obs, (void*)m.acts, (void*)m.done, (void*)m.idx, (void*)m.logp, (void*)m.rew, (void*)m.doneF, (void*)m.trunc, (void*)m.adv, (void*)m.tgt,
                    (void*)m.ret, (void*)m.vals, (void*)m.metrics, (void*)m.scratch, (void*)m.exObs, (void*)m.exActs, (void*)m.exLogp, (void*)m.exAdv, (void*)m.exTgt})
        if (p) (void)hipFree(p);
    if (m.learnPending) (void)hipEventSynchronize(m.evLearnDone);
    if (m.learnStream) { (void)hipStreamDestroy(m.learnStream); (void)hipEventDestroy(m.evReady); (void)hipEventDestroy(m.evLearnDone); }
    if (m.drawPending) m.nextDraw.wait();
    if (m.permPending) m.permDraw.wait();
    for (void* p : {(void*)m.steps, (void*)m.trajOff, (void*)m.permDev}) if (p) (void)hipFree(p);
    delete skillTracker; delete metricSender; delete renderSender;
    delete ppo; delete agentMgr; delete expBuffer;
    if (m.fifo) rlgpu_expbuf_destroy(m.fifo);
    if (m.shuf) rlgpu_shuffler_destroy(m.shuf);
    if (m.lrn) rlgpu_learner_destroy(m.lrn);
    if (m.env) rlgpu_env_destroy(m.env);
    if (m.retShare) (void)hipFree(m.retShare);
    if (m.comm) rlgpu_comm_destroy(m.comm);
    // (the per-env plugin sets, scratch arenas and host-path device rows go with m.hp)
    delete m.gym; delete m.match;   // GameInst deletes its gym and match in the reference (GameInst.h:53-56); plugins stay the user's
    delete impl;
}

int Learner::NumEnvs() const { return impl->nEnvs; }
int Learner::NumAgents() const { return impl->nAgents; }
int Learner::StepsPerIteration() const { return impl->T; }
uint32_t Learner::SamplerCalls() const {
    uint32_t stream = 0, calls = 0;
    impl->LrnCheck(rlgpu_learner_get_sampler(impl->lrn, &stream, &calls), "learner_get_sampler");
    return calls;
}
uint32_t Learner::SamplerStream() const {
    uint32_t stream = 0, calls = 0;
    impl->LrnCheck(rlgpu_learner_get_sampler(impl->lrn, &stream, &calls), "learner_get_sampler");
    return stream;
}
void Learner::CopyCollected(std::vector<float>* obs, std::vector<int32_t>* actions, std::vector<float>* rewards, std::vector<int32_t>* dones) {
    Impl& m = *impl;
    const size_t TN = (size_t)m.Tcap * m.nAgents;
    HOST_HIP(hipDeviceSynchronize());
    if (obs) { obs->resize((size_t)(m.Tcap + 1) * m.nAgents * m.D); HOST_HIP(hipMemcpy(obs->data(), m.obs, obs->size() * 4, hipMemcpyDeviceToHost)); }
    if (actions) { actions->resize(TN); HOST_HIP(hipMemcpy(actions->data(), m.acts, TN * 4, hipMemcpyDeviceToHost)); }
    if (rewards) { rewards->resize(TN); HOST_HIP(hipMemcpy(rewards->data(), m.rew, TN * 4, hipMemcpyDeviceToHost)); }
    if (dones) { dones->resize(TN); HOST_HIP(hipMemcpy(dones->data(), m.done, TN * 4, hipMemcpyDeviceToHost)); }
}
int Learner::StepCapacity() const { return impl->Tcap; }
std::vector<int32_t> Learner::CollectedSteps() const { return impl->hSteps; }
bool Learner::UsesFreeRunningCollection() const { return impl->lastFree; }
uint64_t Learner::LastIterationTimesteps() const { return (uint64_t)impl->lastRows; }
// multi-GPU fail-fast: an error of the ranks' exchange (RCCL's asynchronous error, a lost peer) or replicas that are no longer equal END the
// process with the text -- the launcher restarts every rank from the last checkpoint (a fresh process each: nothing is re-executed in place)
void Learner::CheckReplicas() {
    Impl& m = *impl;
    if (!m.comm) return;
    if (rlgpu_comm_check(m.comm) != RLGPU_OK) RG_ERR_CLOSE("multi-GPU exchange failed: " << rlgpu_comm_last_error(m.comm));
    static const int every = [] { const char* s = std::getenv("RLGPU_REPLICA_CHECK_EVERY"); return s && std::atoi(s) > 0 ? std::atoi(s) : 50; }();
    if (totalIterations % (uint64_t)every != 0) return;
    if (m.learnPending) return;   // (collectionDuringLearn: the epochs in flight are writing the parameters)
    int equal = 1;
    m.LrnCheck(rlgpu_learner_replicas_equal(m.lrn, m.comm, &equal), "learner_replicas_equal");
    if (!equal) RG_ERR_CLOSE("rank " << m.rank << ": parameters differ from rank 0's after " << totalIterations << " iterations (replicas diverged)");
}
uint64_t Learner::ParamChecksum() const {
    uint64_t v = 0;
    impl->LrnCheck(rlgpu_learner_param_checksum(impl->lrn, &v), "learner_param_checksum");
    return v;
}
int Learner::Rank() const { return impl->rank; }
int Learner::WorldSize() const { return impl->world; }
std::vector<double> Learner::GatherOverRanks(double v) {
    Impl& m = *impl;
    if (!m.comm) return {v};
    std::vector<float> h(m.world, 0.f); h[m.rank] = (float)v;      // a sum of one-hot vectors = a gather
    float* d = dev_alloc<float>(m.world);
    HOST_HIP(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    if (rlgpu_comm_allreduce_f32(m.comm, d, m.world, nullptr) != RLGPU_OK) RG_ERR_CLOSE("rlgpu_comm_allreduce_f32: " << rlgpu_comm_last_error(m.comm));
    HOST_HIP(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    dev_release(d);
    return std::vector<double>(h.begin(), h.end());
}
double Learner::MaxOverRanks(double v) {
    const std::vector<double> all = GatherOverRanks(v);
    return *std::max_element(all.begin(), all.end());
}
void Learner::AllReduceTimings(float& ms, int& calls, bool reset) {
    Impl& m = *impl;
    m.arTimed = true;
    for (size_t i = 0; i < m.arUsed; i++) {
        float t = 0.f;
        HOST_HIP(hipEventSynchronize(m.arEvents[i].second));
        HOST_HIP(hipEventElapsedTime(&t, m.arEvents[i].first, m.arEvents[i].second));
        m.arMs += t; m.arCalls++;
    }
    m.arUsed = 0;
    ms = (float)m.arMs; calls = m.arCalls;
    if (reset) { m.arMs = 0; m.arCalls = 0; }
}
// (RLGPU_FUSED_MAX_TEAM = 2 puts 3v3 back on alternating act / step launches: the faster way while a wavefront held ONE 3v3 env)
static int FusedMaxTeam() { static const int v = [] { const char* s = std::getenv("RLGPU_FUSED_MAX_TEAM"); return s ? std::atoi(s) : 3; }(); return v; }
bool Learner::UsesFusedCollection() const { return impl->fusedCollect && impl->match->teamSize <= FusedMaxTeam() && (!(stepCallback || impl->plan.AnyHost()) || impl->DeferredNow(this)) && !renderSender; }
void Learner::DeviceTimings(float& envMs, int& envLaunches, float& gemmMs, double& gemmFlops, int& gemmCalls, bool reset) {
    // the device-side clocks are opt-in: the first call switches them on (a training run that never asks pays for no events)
    impl->EnvCheck(rlgpu_env_enable_timing(impl->env, 1), "enable_timing");
    impl->LrnCheck(rlgpu_learner_enable_timing(impl->lrn, 1), "learner_enable_timing");
    impl->EnvCheck(rlgpu_env_timing_total(impl->env, &envMs, &envLaunches, reset ? 1 : 0), "timing_total");
    impl->LrnCheck(rlgpu_learner_timing_total(impl->lrn, &gemmMs, &gemmFlops, &gemmCalls, reset ? 1 : 0), "learner_timing_total");
}

void Learner::UpdateLearningRates(float policyLR, float criticLR) {
    config.ppo.policyLR = policyLR; config.ppo.criticLR = criticLR;
    impl->LrnCheck(rlgpu_learner_set_lr(impl->lrn, policyLR, criticLR), "learner_set_lr");
}

// ThreadAgent::_RunFunc (ThreadAgent.cpp:24-195) for every game at once: policy inference and the env step never leave the device
// ---- the host path --------------------------------------------------------------------------------------------------------------
void Learner::Impl::SetupHostPath(const EnvCreateFn& create, int numThreads) {
    if (hp.ready) return;
    hp.allocF32 = [](size_t n) { return dev_alloc<float>(n); };
    hp.Setup(env, plan, match, gym, create, numThreads, nEnvs, nPlayers, nAgents, D, Ddev, tickSkip);
    games.resize(nEnvs);
    for (int e = 0; e < nEnvs; e++) { games[e].gym = hp.envGym[e]; games[e].match = hp.envMatch[e]; games[e].index = e; }
}

// One step of every game with host work in it: the policy's actions are on the device already (acts + t * nAgents).  HostEnvPath::Step does
// Gym::Step's part (Gym.cpp:68-102); GameInst::Step's bookkeeping (GameInst.cpp:7-38) and the step callback run per env here.
void Learner::Impl::HostStep(Learner* self, int t) {
    const size_t o = (size_t)t * nAgents;
    const int P = nPlayers;
    const StepCallback& callback = self->stepCallback;
    hp.D = D;   // (the obs width of a host builder is known only after the first reset: Learner's constructor)
    hp.Step(acts + o, ObsAt(t + 1), rew + o, done + o, [&](int e, RLGSC::Gym::StepResult& sr) {
        GameInst& g = games[e];   // GameInst.cpp:14-34
        const float sum = std::accumulate(sr.reward.begin(), sr.reward.end(), 0.f);
        g.avgStepRew.Add(sum, (uint64_t)P); g.curEpRew += sum / P; g.totalSteps++;
        if (callback) callback(&g, sr, g._metrics);
        if (sr.done) { g.avgEpRew += g.curEpRew; g.curEpRew = 0; }
    });
}

bool Learner::Impl::DeferredNow(const Learner* self) const {
    if (!self->config.deferHostRewards || !fusedCollect || match->teamSize > FusedMaxTeam() || self->renderSender || !match->spawnOpponents) return false;
    return deferredReward || (!plan.AnyHost() && (bool)self->stepCallback);
}
// the steps of the launch that just ended, per env in order: rewards (when the reward function is the user's), GameInst's bookkeeping (GameInst.cpp:14-34), the callback
void Learner::Impl::DeferredReplay(Learner* self) {
    const StepCallback& callback = self->stepCallback;
    const int P = nPlayers;
    hp.D = D;
    hp.ReplayCollected(Tused, hSteps.data(), acts, rew, done, [&](int e, RLGSC::Gym::StepResult& sr) {
        GameInst& g = games[e];
        const float sum = std::accumulate(sr.reward.begin(), sr.reward.end(), 0.f);
        g.avgStepRew.Add(sum, (uint64_t)P); g.curEpRew += sum / P; g.totalSteps++;
        if (callback) callback(&g, sr, g._metrics);
        if (sr.done) { g.avgEpRew += g.curEpRew; g.curEpRew = 0; }
    });
}

void Learner::CollectTimesteps() {
    Impl& m = *impl;
    const size_t rowObs = (size_t)m.nAgents * m.D;
    if (!m.first) {
        // the observation every player acts on next = the row after its own last step
        if (m.ragged) hipLaunchKernelGGL(k_carry_obs, dim3((unsigned)m.nAgents), dim3(64), 0, nullptr, m.obs, (const int32_t*)m.steps, m.nPlayers, m.nAgents, m.D);
        else HOST_HIP(hipMemcpyAsync(m.ObsAt(0), m.ObsAt(m.T), rowObs * 4, hipMemcpyDeviceToDevice, nullptr));
    }
    m.first = false;
    const bool hostWork = (bool)stepCallback || m.plan.AnyHost();
    const bool deferred = hostWork && m.DeferredNow(this);   // a user reward / a callback and nothing else: replayed after the fused launch
    const bool slow = hostWork && !deferred;
    if (hostWork && !m.hp.ready) {   // a step callback was installed after construction
        m.SetupHostPath(envCreateFn, config.numThreads);
        m.EnvCheck(rlgpu_env_download_states(m.env, m.hp.snaps.data(), nullptr, m.nEnvs), "download_states");
        for (int e = 0; e < m.nEnvs; e++) m.hp.prevGs[e] = RLGSC::GameState(m.hp.snaps[e], m.tickSkip);
    }
    if (deferred && m.hp.recCap < m.Tcap) m.hp.EnableStepRecords(m.Tcap);
    if (!deferred && m.hp.recCap > 0) m.hp.EnableStepRecords(0);   // (the callback went away)
    int64_t lockstepJobRows = -1;   // (set when the other ranks of this iteration collected free-running: see below)
    auto lockstepDone = [&]() {   // every game made T steps
        if (m.ragged) hipLaunchKernelGGL(k_fill_i32, dim3((unsigned)((m.nEnvs + 255) / 256)), dim3(256), 0, nullptr, m.steps, m.nEnvs, (int32_t)m.T);
        std::fill(m.hSteps.begin(), m.hSteps.end(), m.T); std::fill(m.hAgentSteps.begin(), m.hAgentSteps.end(), m.T);
        m.Tused = m.T; m.lastRows = m.B; m.lastFree = false; m.lastRowsAll = lockstepJobRows >= 0 ? lockstepJobRows : (int64_t)m.B * m.world;
        totalTimesteps += (uint64_t)m.lastRowsAll;
    };
    // no per-step host work: the whole phase in one launch (rlgpu_collect / rlgpu_collect_free), when the policy fits the in-kernel inference
    if (!slow && !renderSender && m.fusedCollect && m.match->teamSize <= FusedMaxTeam()) {
        if (m.ragged && m.freeOk) {
            // the agents run free until the batch has its timesteps together (ThreadAgentManager.cpp:16-32)
            int rc = rlgpu_collect_free(m.env, m.lrn, m.Tcap, m.B, m.obs, m.acts, m.logp, m.rew, m.done, m.steps, config.deterministic ? 1 : 0);
            if (rc != RLGPU_OK && rc != RLGPU_ERR_STATE) m.EnvCheck(rc, "collect_free");
            int64_t rows = 0; int tmax = 0;
            if (rc == RLGPU_OK) {
                HOST_HIP(hipMemcpy(m.hSteps.data(), m.steps, (size_t)m.nEnvs * 4, hipMemcpyDeviceToHost));   // (waits for the launch)
                for (int e = 0; e < m.nEnvs; e++) {
                    const int32_t st = m.hSteps[(size_t)e];
                    rows += (int64_t)st * m.nPlayers; tmax = std::max(tmax, (int)st);
                    for (int k = 0; k < m.nPlayers; k++) m.hAgentSteps[(size_t)e * m.nPlayers + k] = st;
                }
            }
            // every rank's own pace gives it its own count (they differ by less than one step of every game): the job's count is the SUM, taken
            // collectively so that totalTimesteps -- the Learn() loop's exit, the checkpoint cadence and folder names -- is the same number on every rank.
            // The same exchange carries each rank's verdict on free-running itself (-1: its batch is not resident at once, or the policy does not fit the
            // kernel): EVERY rank makes this call in this branch, so a rank that has to fall back never leaves the others waiting in a collective, and
            // from the next iteration on all of them collect in lockstep.
            int64_t jobRows = rc == RLGPU_OK ? rows : m.B; bool allFree = rc == RLGPU_OK;
            if (m.comm) {
                jobRows = 0;
                for (double v : GatherOverRanks(rc == RLGPU_OK ? (double)rows : -1.0)) { if (v < 0) { allFree = false; jobRows += m.B; } else jobRows += (int64_t)std::llround(v); }
            }
            if (!allFree) {
                m.freeOk = false;
                if (m.rank == 0 && !std::getenv("RLGPU_QUIET")) RG_LOG("Learner: " << (rc == RLGPU_OK ? "another rank cannot collect free-running" : rlgpu_env_last_error(m.env)) << " -> lockstep collection");
            }
            if (rc == RLGPU_OK) {
                m.Tused = tmax; m.lastRows = rows; m.lastFree = true;
                m.lastRowsAll = jobRows;
                totalTimesteps += (uint64_t)m.lastRowsAll;
                if (deferred) m.DeferredReplay(this);
                return;
            }
            lockstepJobRows = jobRows;   // this rank collects the iteration in lockstep below; the job's count is the one every rank computed
        }
        int rc = rlgpu_collect(m.env, m.lrn, m.T, m.obs, m.acts, m.logp, m.rew, m.done, config.deterministic ? 1 : 0);
        if (rc == RLGPU_OK) { lockstepDone(); if (deferred) m.DeferredReplay(this); return; }
        if (rc != RLGPU_ERR_STATE) m.EnvCheck(rc, "collect");
        m.fusedCollect = false;   // fp32 mode or a net too wide for the kernel's LDS scratch: alternate act / step
    }
    for (int t = 0; t < m.T; t++) {
        const size_t o = (size_t)t * m.nAgents;
        m.LrnCheck(rlgpu_policy_act(m.lrn, m.ObsAt(t), m.nAgents, config.deterministic ? 1 : 0, nullptr, m.acts + o, m.logp + o), "policy_act");
        if (hostWork) m.HostStep(this, t);   // (also a deferred kind whose fused launch is not available: fp32 mode, wide nets)
        else m.EnvCheck(rlgpu_env_step(m.env, m.acts + o, m.ObsAt(t + 1), m.rew + o, m.done + o), "step");
        if (renderSender) RenderStep(t);
    }
    lockstepDone();
}

// ThreadAgent.cpp:164-186: the first game's state goes to the renderer after every step, paced to renderTimeScale x real time
void Learner::RenderStep(int t) {
    Impl& m = *impl;
    RlgpuArenaState st; const int32_t env0 = 0;
    std::vector<int32_t> acts(m.nPlayers);
    m.EnvCheck(rlgpu_env_download_states(m.env, &st, &env0, 1), "download_states");
    HOST_HIP(hipMemcpy(acts.data(), m.acts + (size_t)t * m.nAgents, (size_t)m.nPlayers * 4, hipMemcpyDeviceToHost));
    RLGSC::GameState gs(st, m.tickSkip);
    renderSender->Send(gs, m.match->actionParser->ParseActions(RLGSC::IList(acts.begin(), acts.end()), gs));
    const double target = (1 / 120.0) * m.tickSkip / std::max(config.renderTimeScale, 1e-3f);
    const double sleepTime = std::max(target - m.renderTimer.Elapsed(), 0.0);
    if (!std::getenv("RLGPU_RENDER_NO_SLEEP")) std::this_thread::sleep_for(std::chrono::microseconds((int64_t)(sleepTime * 1e6)));
    m.renderTimer.Reset();
}

std::vector<Report> Learner::GetAllGameMetrics() {
    std::vector<Report> out;
    for (auto& g : impl->games) { out.push_back(g._metrics); g.ResetMetrics(); }
    return out;
}

// Learner.cpp:608-703 for an iteration whose trajectories have their own lengths (steps[e] rows for the players of game e, laid out time-major
// with Tcap rows of room): the batch is the concatenation of the trajectories, as ThreadAgentManager::CollectTimesteps hands it over
void Learner::AddNewExperienceRagged(Report& report) {
    Impl& m = *impl;
    const size_t TU = (size_t)m.Tused * m.nAgents, rows = TU + m.nAgents, slotRows = (size_t)m.Tcap * m.nAgents;
    const int64_t R = m.lastRows;
    for (size_t s = 0; s < rows; s += m.maxRows) {   // value predictions of every state incl. the one after each trajectory's last step (rows beyond a trajectory are not used)
        int n = (int)std::min<size_t>(m.maxRows, rows - s);
        m.LrnCheck(rlgpu_value_forward(m.lrn, m.obs + s * m.D, n, m.vals + s), "value_forward");
    }
    const float retStd = config.standardizeReturns ? (float)returnStats.GetSTD() : 1.f;   // read BEFORE this batch updates it (Learner.cpp:651)
    if (TU) hipLaunchKernelGGL(k_done_f, dim3((unsigned)((TU + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)m.done, TU, m.doneF);
    // truncation marks: the last step of every trajectory unless done (ThreadAgentManager.cpp:55) -- applied by the kernel
    m.LrnCheck(rlgpu_gae_ragged(m.lrn, m.rew, m.doneF, nullptr, m.vals, m.nAgents, m.steps, m.nPlayers, config.gaeGamma, config.gaeLambda, retStd, config.rewardClipRange,
                                config.gaeNextValueMode, m.adv, m.tgt, m.ret), "gae_ragged");
    if (config.standardizeReturns) {   // the first <= maxReturnsPerStatsInc returns of the concatenated batch (Learner.cpp:679-682): trajectory 0's, then trajectory 1's, ...
        const int k = (int)std::min<int64_t>(config.maxReturnsPerStatsInc, R);
        for (int j = 0, done_k = 0; done_k < k && j < m.nAgents; j++) {
            const int cnt = std::min((int)m.hAgentSteps[(size_t)j], k - done_k);
            if (cnt > 0) HOST_HIP(hipMemcpy2DAsync(m.retShare + done_k, 4, m.ret + j, (size_t)m.nAgents * 4, 4, cnt, hipMemcpyDeviceToDevice, nullptr));
            done_k += cnt;
        }
        if (m.comm && rlgpu_comm_broadcast(m.comm, m.retShare, (int64_t)k * 4, 0, nullptr) != RLGPU_OK) RG_ERR_CLOSE("rlgpu_comm_broadcast: " << rlgpu_comm_last_error(m.comm));
        FList first(k);
        HOST_HIP(hipMemcpy(first.data(), m.retShare, (size_t)k * 4, hipMemcpyDeviceToHost));
        returnStats.Increment(first, k);
    }
    HOST_HIP(hipMemsetAsync(m.scratch, 0, 32, nullptr));
    hipLaunchKernelGGL(k_ragged_sums, dim3(128), dim3(256), 0, nullptr, (const float*)m.ret, (const float*)m.adv, (const float*)m.tgt, (const float*)m.rew, (const int32_t*)m.steps,
                       m.nPlayers, m.Tused, m.nAgents, m.scratch);
    float h[4];
    HOST_HIP(hipMemcpy(h, m.scratch, 16, hipMemcpyDeviceToHost));
    const float inv = 1.f / (float)std::max<int64_t>(R, 1);
    report["Avg Return"] = h[0] * inv / retStd; report["Avg Advantage"] = h[1] * inv; report["Avg Val Target"] = h[2] * inv;
    report["Average Step Reward"] = h[3] * inv;
    // ExperienceBuffer::SubmitExperience (Learner.cpp:694-702).  On a multi-GPU run only the batch's last B rows join, so that every rank's FIFO
    // holds the same number of rows and the ranks make the same number of optimizer steps (one all-reduce each)
    int slot = -1;
    if (rlgpu_expbuf_submit_ragged(m.fifo, m.hAgentSteps.data(), m.world > 1 ? m.B : 0, &slot) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: no free slot");
    if (m.learnPending) HOST_HIP(hipStreamWaitEvent(nullptr, m.evLearnDone, 0));   // collectionDuringLearn: the epochs still running read the slot this may overwrite
    if (rlgpu_traj_offsets(m.steps, m.nAgents, m.nPlayers, m.trajOff + (size_t)slot * (m.nAgents + 1), nullptr) != RLGPU_OK) RG_ERR_CLOSE("rlgpu_traj_offsets failed");
    const size_t o = (size_t)slot * slotRows;
    if (TU) {
        HOST_HIP(hipMemcpyAsync(m.exObs + o * m.D, m.obs, TU * m.D * 4, hipMemcpyDeviceToDevice, nullptr));
        HOST_HIP(hipMemcpyAsync(m.exActs + o, m.acts, TU * 4, hipMemcpyDeviceToDevice, nullptr));
        HOST_HIP(hipMemcpyAsync(m.exLogp + o, m.logp, TU * 4, hipMemcpyDeviceToDevice, nullptr));
        HOST_HIP(hipMemcpyAsync(m.exAdv + o, m.adv, TU * 4, hipMemcpyDeviceToDevice, nullptr));
        HOST_HIP(hipMemcpyAsync(m.exTgt + o, m.tgt, TU * 4, hipMemcpyDeviceToDevice, nullptr));
    }
}

void Learner::AddNewExperience(Report& report) {
    Impl& m = *impl;
    if (m.ragged) { AddNewExperienceRagged(report); return; }
    const size_t TN = (size_t)m.T * m.nAgents, rows = TN + m.nAgents;
    for (size_t s = 0; s < rows; s += m.maxRows) {   // minibatched value predictions, incl. the states after the last step (Learner.cpp:619-640)
        int n = (int)std::min<size_t>(m.maxRows, rows - s);
        m.LrnCheck(rlgpu_value_forward(m.lrn, m.obs + s * m.D, n, m.vals + s), "value_forward");
    }
    const float retStd = config.standardizeReturns ? (float)returnStats.GetSTD() : 1.f;   // read BEFORE this batch updates it (Learner.cpp:651)
    hipLaunchKernelGGL(k_done_trunc, dim3((unsigned)((TN + 255) / 256)), dim3(256), 0, nullptr, (const int32_t*)m.done, m.T, m.nAgents, m.doneF, m.trunc);
    m.LrnCheck(rlgpu_gae(m.lrn, m.rew, m.doneF, m.trunc, m.vals, m.T, m.nAgents, config.gaeGamma, config.gaeLambda, retStd, config.rewardClipRange, config.gaeNextValueMode, m.adv, m.tgt, m.ret), "gae");
    if (config.standardizeReturns) {
        // the first <= maxReturnsPerStatsInc returns of the concatenated batch (Learner.cpp:679-682), which is agent-major: trajectory 0's
        // T returns, then trajectory 1's, ... -- gathered column by column from the time-major device array; on a multi-GPU run every
        // rank feeds rank 0's, so the statistic (and with it retStd) is identical everywhere
        const int k = (int)std::min<int64_t>(config.maxReturnsPerStatsInc, (int64_t)m.T * m.nAgents);
        for (int j = 0, done_k = 0; done_k < k; j++) {
            const int cnt = std::min(m.T, k - done_k);
            HOST_HIP(hipMemcpy2DAsync(m.retShare + done_k, 4, m.ret + j, (size_t)m.nAgents * 4, 4, cnt, hipMemcpyDeviceToDevice, nullptr));
            done_k += cnt;
        }
        if (m.comm && rlgpu_comm_broadcast(m.comm, m.retShare, (int64_t)k * 4, 0, nullptr) != RLGPU_OK) RG_ERR_CLOSE("rlgpu_comm_broadcast: " << rlgpu_comm_last_error(m.comm));
        FList first(k);
        HOST_HIP(hipMemcpy(first.data(), m.retShare, (size_t)k * 4, hipMemcpyDeviceToHost));
        returnStats.Increment(first, k);
    }
    float sums[3] = {0, 0, 0};
    HOST_HIP(hipMemsetAsync(m.scratch, 0, 32, nullptr));
    hipLaunchKernelGGL(k_abs_sums, dim3(256), dim3(256), 0, nullptr, (const float*)m.ret, (const float*)m.adv, (const float*)m.tgt, TN, m.scratch);
    hipLaunchKernelGGL(k_sum, dim3(256), dim3(256), 0, nullptr, (const float*)m.rew, TN, m.scratch + 3);
    float h[4];
    HOST_HIP(hipMemcpy(h, m.scratch, 16, hipMemcpyDeviceToHost));
    for (int i = 0; i < 3; i++) sums[i] = h[i] / (float)TN;
    report["Avg Return"] = sums[0] / retStd; report["Avg Advantage"] = sums[1]; report["Avg Val Target"] = sums[2];
    report["Average Step Reward"] = h[3] / (float)TN;
    // ExperienceBuffer::SubmitExperience (Learner.cpp:694-702): this iteration's rows join the FIFO
    int slot = m.pendingSlot;   // LearnPPO already accounted for this iteration when it drew the permutation ahead
    m.pendingSlot = -1;
    if (slot < 0 && rlgpu_expbuf_submit(m.fifo, &slot) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: no free slot");
    const size_t o = (size_t)slot * TN;
    HOST_HIP(hipMemcpyAsync(m.exObs + o * m.D, m.obs, TN * m.D * 4, hipMemcpyDeviceToDevice, nullptr));
    HOST_HIP(hipMemcpyAsync(m.exActs + o, m.acts, TN * 4, hipMemcpyDeviceToDevice, nullptr));
    HOST_HIP(hipMemcpyAsync(m.exLogp + o, m.logp, TN * 4, hipMemcpyDeviceToDevice, nullptr));
    HOST_HIP(hipMemcpyAsync(m.exAdv + o, m.adv, TN * 4, hipMemcpyDeviceToDevice, nullptr));
    HOST_HIP(hipMemcpyAsync(m.exTgt + o, m.tgt, TN * 4, hipMemcpyDeviceToDevice, nullptr));
}

void Learner::LearnPPO(Report& report) {
    Impl& m = *impl;
    if (config.deterministic) RG_ERR_CLOSE("PPOLearner::Learn() called with config.deterministic = true");   // Learner.cpp:472-477
    const bool overlap = m.learnStream != nullptr;
    hipStream_t ls = overlap ? m.learnStream : nullptr;
    if (overlap) {   // this iteration's rows are in their FIFO slot once the default stream gets here
        HOST_HIP(hipEventRecord(m.evReady, nullptr)); HOST_HIP(hipStreamWaitEvent(ls, m.evReady, 0));
        m.LrnCheck(rlgpu_learner_set_stream(m.lrn, (void*)ls), "learner_set_stream");
    }
    HOST_HIP(hipMemsetAsync(m.metrics, 0, 32, ls));
    int nMini = 0, nUpdates = 0;
    Timer t;
    for (int ep = 0; ep < config.ppo.epochs; ep++) {
        // ExperienceBuffer::GetAllBatchesShuffled (ExperienceBuffer.cpp:104-126) over the whole FIFO: logical rows are oldest iteration
        // first and agent-major inside one (trajectory after trajectory); the device slots are time-major
        int64_t cur;
        if (m.ragged) {
            // the permutation depends on the FIFO's SIZE only: it was drawn (by a worker, beside the GPU's work) for the size expected; the rows it
            // stands for depend on this iteration's trajectory lengths and are looked up on the device
            cur = rlgpu_expbuf_size(m.fifo);
            const int32_t* perm = m.TakePerm(cur);
            HOST_HIP(hipMemcpyAsync(m.permDev, perm, (size_t)cur * 4, hipMemcpyHostToDevice, ls));   // pageable source: staged before the call returns
            if (rlgpu_expbuf_map_rows_dev(m.fifo, m.permDev, cur, m.trajOff, m.idx, (void*)ls) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: row mapping failed");
            // next: this FIFO again (another epoch), or the FIFO after the next iteration joined (as many rows as this one brought, or the buffer full)
            const int64_t add = m.world > 1 ? std::min<int64_t>(m.lastRows, m.B) : m.lastRows;
            m.StartPermDraw(ep == config.ppo.epochs - 1 ? std::min<int64_t>(cur + std::min<int64_t>(add, config.expBufferSize), config.expBufferSize) : cur);
        } else {
        cur = m.TakeDraw();
        HOST_HIP(hipMemcpyAsync(m.idx, m.phys[m.physFlip].data(), (size_t)cur * 4, hipMemcpyHostToDevice, ls));   // pageable source: staged before the call returns
        m.physFlip ^= 1;
        if (ep == config.ppo.epochs - 1 && rlgpu_expbuf_submit(m.fifo, &m.pendingSlot) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: no free slot");   // the next draw sees the FIFO after the next submit
        m.StartDraw();
        }
        for (int64_t b = 0; b + m.batch <= cur; b += m.batch) {   // the remainder is dropped (ExperienceBuffer.cpp:115-117)
            m.LrnCheck(rlgpu_zero_grads(m.lrn), "zero_grads");
            for (int64_t k = 0; k < m.batch; k += m.mini) {
                m.LrnCheck(rlgpu_ppo_minibatch(m.lrn, m.exObs, m.exActs, m.exLogp, m.exAdv, m.exTgt, m.idx + b + k, (int)m.mini, (float)m.mini / (float)m.batch, m.metrics), "ppo_minibatch");
                nMini++;
            }
            // multi-GPU: ONE all-reduce(sum) of the flat [policy | critic] gradient on the learner's stream, then scale by 1 / world INSIDE
            // the clip so the norm is taken of the global-batch gradient, like a single learner on the union would
            if (m.comm) {
                std::pair<hipEvent_t, hipEvent_t>* ev = nullptr;
                if (m.arTimed) {   // (bench only: a pair of events around the collective on the learner's stream)
                    if (m.arUsed == m.arEvents.size()) {
                        if (m.arEvents.size() < 1024) { hipEvent_t a, b; HOST_HIP(hipEventCreate(&a)); HOST_HIP(hipEventCreate(&b)); m.arEvents.push_back({a, b}); }
                        else { float t; int c; AllReduceTimings(t, c, false); }
                    }
                    ev = &m.arEvents[m.arUsed++];
                    HOST_HIP(hipEventRecord(ev->first, ls));
                }
                m.LrnCheck(rlgpu_allreduce_grads(m.lrn, m.comm), "allreduce_grads");
                if (ev) HOST_HIP(hipEventRecord(ev->second, ls));
            }
            m.LrnCheck(rlgpu_clip_adam_step(m.lrn, 0.5f, 1.f / (float)m.world), "clip_adam_step");   // clip_grad_norm_(0.5) per network, then Adam (PPOLearner.cpp:273-288)
            if (overlap) m.LrnCheck(rlgpu_learner_refresh_shadows(m.lrn), "learner_refresh_shadows");   // the collector reads the bf16 copies while this stream goes on: keep them live
            nUpdates++;
        }
    }
    totalEpochs += config.ppo.epochs; m.cumulativeModelUpdates += nUpdates;
    report["Cumulative Model Updates"] = (double)m.cumulativeModelUpdates;
    if (overlap) {   // the epochs go on while the next iteration is collected; FinishLearn() reads their statistics then
        HOST_HIP(hipEventRecord(m.evLearnDone, ls));
        m.LrnCheck(rlgpu_learner_set_stream(m.lrn, nullptr), "learner_set_stream");
        m.learnPending = true; m.pendMini = nMini; m.pendTimer = t;
        return;
    }
    m.LrnCheck(rlgpu_learner_sync(m.lrn), "learner_sync");
    m.pendMini = nMini; m.pendTimer = t;
    FinishLearn(report);
}

// the PPO statistics of the epochs LearnPPO launched (with collectionDuringLearn: one iteration later, when they are done)
void Learner::FinishLearn(Report& report) {
    Impl& m = *impl;
    if (m.learnPending) { HOST_HIP(hipEventSynchronize(m.evLearnDone)); m.learnPending = false; }
    float h[8];
    HOST_HIP(hipMemcpy(h, m.metrics, 32, hipMemcpyDeviceToHost));
    const double rows = std::max<double>(1, (double)m.pendMini * m.mini);
    report["Policy Entropy"] = h[0] / rows; report["Mean KL Divergence"] = h[1] / rows; report["SB3 Clip Fraction"] = h[2] / rows;
    report["Value Function Loss"] = h[4] / rows; report["PPO Learn Time"] = m.pendTimer.Elapsed();
}


void Learner::Learn() {
    Impl& m = *impl;
    RG_LOG("Learner: " << m.nEnvs << " envs (" << m.nAgents << " agents), obs " << m.D << ", actions " << m.A << ", " << m.T << " steps/env/iteration = "
           << m.B << " timesteps" << (m.ragged ? " (free-running collection: every game at its own pace until the batch has them)" : "") << ", batch " << m.batch << ", minibatch " << m.mini);
    while (config.timestepLimit == 0 || totalTimesteps < config.timestepLimit) {
        if (m.renderOnly) { CollectTimesteps(); continue; }   // render mode: play the (loaded) policy forever, one step per "iteration"
        Report report;
        Timer tAll, tCollect;
        CollectTimesteps();
        HOST_HIP(hipDeviceSynchronize());
        double collectTime = tCollect.Elapsed();
        if (m.learnPending) FinishLearn(report);   // collectionDuringLearn: the previous iteration's epochs ran beside this collection
        if (config.deviceStepMetrics) {   // what examplemain.cpp's step callback averages, from the device's running totals of this iteration
            float st[4];
            m.EnvCheck(rlgpu_env_step_stats(m.env, st, 1), "step_stats");
            const double n = std::max(1.f, st[0]);
            report["player_speed"] = st[1] / n; report["ball_touch_ratio"] = st[2] / n; report["in_air_ratio"] = st[3] / n;
        }
        Timer tConsume;
        AddNewExperience(report);
        LearnPPO(report);
        double consumeTime = tConsume.Elapsed();
        if (skillTracker) {   // Learner.cpp:527-538
            RG_LOG("Running skill eval game(s)...");
            if (config.skillTrackerConfig.stepCallback == NULL) skillTracker->config.stepCallback = stepCallback;
            skillTracker->RunGames((int64_t)m.lastRows);
            for (auto& pair : skillTracker->curRating.data) report[std::string("Skill Rating") + (pair.first.empty() ? "" : " ") + pair.first] = pair.second;
        }
        totalIterations++;
        CheckReplicas();
        report["Total Iterations"] = (double)totalIterations; report["Cumulative Timesteps"] = (double)totalTimesteps;
        report["Timesteps Collected"] = (double)m.lastRows;
        report["Collection Time"] = collectTime; report["Consumption Time"] = consumeTime; report["Total Iteration Time"] = tAll.Elapsed();
        report["Collected Steps/Second"] = (double)m.lastRowsAll / std::max(collectTime, 1e-9);
        report["Overall Steps/Second"] = (double)m.lastRowsAll / std::max(tAll.Elapsed(), 1e-9);
        if (iterationCallback) iterationCallback(this, report);
        if (config.sendMetrics && metricSender) metricSender->Send(report);                                                                       // Learner.cpp:589-590
        if (m.rank != 0 || std::getenv("RLGPU_QUIET")) { m.tsSinceSave += (uint64_t)m.lastRowsAll; if (m.rank == 0 && !config.checkpointSaveFolder.empty() && m.tsSinceSave >= (uint64_t)std::max<int64_t>(config.timestepsPerSave, 1)) Save(); continue; }
        RG_LOG(std::string(8, '\n') << std::string(20, '=') << " ITERATION COMPLETED " << std::string(20, '='));
        const std::vector<std::string> rows = {"Average Step Reward", "Policy Entropy", "Value Function Loss", "", "Mean KL Divergence", "SB3 Clip Fraction", "Avg Return",
                        "Avg Advantage", "Avg Val Target", "", "Collected Steps/Second", "Overall Steps/Second", "", "Collection Time", "Consumption Time",
                        "-PPO Learn Time", "Total Iteration Time", "", "Cumulative Model Updates", "Cumulative Timesteps", "Total Iterations"};
        report.Display(rows);
        // metrics the iteration callback added go where the reference sends them to its metrics receiver: here, the log
        for (auto& kv : report.data) {
            std::string dashed = "-" + kv.first;
            if (std::find(rows.begin(), rows.end(), kv.first) == rows.end() && std::find(rows.begin(), rows.end(), dashed) == rows.end()) RG_LOG("  [metric] " << report.SingleToString(kv.first));
        }
        m.tsSinceSave += (uint64_t)m.lastRowsAll;
        if (m.rank == 0 && !config.checkpointSaveFolder.empty() && m.tsSinceSave >= (uint64_t)std::max<int64_t>(config.timestepsPerSave, 1)) Save();
    }
    if (m.learnPending) { Report last; FinishLearn(last); }
    if (m.rank == 0 && !config.checkpointSaveFolder.empty()) Save();   // rank 0 owns the checkpoints
}

// ---- checkpoints ------------------------------------------------------------------------------------------------------
namespace {
// {inputs, hidden..., outputs} of one network, as rlgpu_lt_* wants it
std::vector<int32_t> DimsOf(int in, const IList& hidden, int out) { std::vector<int32_t> d{in}; for (int h : hidden) d.push_back(h); d.push_back(out); return d; }
void LtCheck(int rc, const char* what, const std::filesystem::path& p) { if (rc != RLGPU_OK) RG_ERR_CLOSE(what << " " << p.string() << ": " << rlgpu_lt_last_error()); }
}  // namespace

void Learner::SaveStats(std::filesystem::path path) {
    std::ofstream f(path);
    if (!f.good()) RG_ERR_CLOSE("Learner::SaveStats(): Can't open file at " << path.string());
    // "var" is the raw sum of squared deviations: the reference writes its `runningVariance` member as is (Learner.cpp:196-202)
    f << std::setprecision(17) << "{\n    \"cumulative_model_updates\": " << impl->cumulativeModelUpdates << ",\n    \"cumulative_timesteps\": " << totalTimesteps
      << ",\n    \"epoch\": " << totalEpochs << ",\n    \"sampler_calls\": " << SamplerCalls() << ",\n    \"env_stream_epoch\": " << impl->envStreamEpoch << ",\n    \"reward_running_stats\": {\n        \"count\": " << returnStats.count << ",\n        \"mean\": [\n            "
      << returnStats.runningMean << "\n        ],\n        \"shape\": 1,\n        \"var\": [\n            " << returnStats.runningVariance << "\n        ]\n    }";
    if (skillTracker) f << ",\n    \"skill_rating\": " << skillTracker->RatingsToJSON();                                          // :185-194
    if (config.sendMetrics && metricSender) f << ",\n    \"run_id\": " << MetricSender::JsonString(metricSender->curRunID);   // :204-205
    f << "\n}";
}
void Learner::LoadStats(std::filesystem::path path) {
    std::ifstream f(path);
    if (!f) RG_ERR_CLOSE("Learner::LoadStats(): Can't open file at " << path.string());
    std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    auto at = [&](const std::string& key, bool required) {
        size_t i = s.find("\"" + key + "\"");
        if (i == std::string::npos) { if (required) RG_ERR_CLOSE(path.string() << ": missing \"" << key << "\""); return i; }
        i = s.find(':', i) + 1;
        while (i < s.size() && (s[i] == ' ' || s[i] == '[' || s[i] == '\n' || s[i] == '\r' || s[i] == '\t')) i++;
        return i;
    };
    auto num = [&](const std::string& key) { return std::stod(s.substr(at(key, true))); };
    totalTimesteps = (uint64_t)num("cumulative_timesteps"); impl->cumulativeModelUpdates = (uint64_t)num("cumulative_model_updates"); totalEpochs = (uint64_t)num("epoch");
    returnStats.runningMean = num("mean"); returnStats.count = (int64_t)num("count"); returnStats.runningVariance = num("var");
    // (additions of this build, also written by the Python host) where the action sampler's and the env batch's counter-based RNG streams go on
    {
        const size_t sc = at("sampler_calls", false), ee = at("env_stream_epoch", false);
        const uint32_t calls = sc != std::string::npos ? (uint32_t)std::stod(s.substr(sc)) : (uint32_t)(totalTimesteps / (uint64_t)std::max(1, impl->nAgents));
        impl->LrnCheck(rlgpu_learner_set_sampler(impl->lrn, (uint32_t)impl->rank, calls), "learner_set_sampler");
        impl->envStreamEpoch = (ee != std::string::npos ? (uint32_t)std::stod(s.substr(ee)) : 0u) + 1u;
        impl->EnvCheck(rlgpu_env_reseed(impl->env, (uint32_t)config.randomSeed + 1000u * (uint32_t)impl->rank, impl->envStreamEpoch), "reseed");
        // the constructor reset every env with epoch 0's streams: draw the first states of the resumed run from the new epoch's
        // (otherwise its first episodes would start from the initial states of the run it continues)
        // (with a host OBS builder the device rows go to the scratch buffer -- the constructor's probe did the same: the device setter must run
        // with the new epoch in that case too)
        impl->EnvCheck(rlgpu_env_reset(impl->env, 1, impl->plan.hostObs ? impl->hp.devObs : impl->ObsAt(0)), "reset after reseed");
        if (impl->plan.AnyHost()) {
            std::vector<int32_t> all(impl->nEnvs); std::iota(all.begin(), all.end(), 0);
            impl->hp.ResetEnvs(all, impl->ObsAt(0), true);
        }
    }
    size_t k = at("skill_rating", false);   // Learner.cpp:229-231
    if (skillTracker && k != std::string::npos) skillTracker->curRating = skillTracker->LoadRatingSet(s[k] == '{' ? s.substr(k, s.find('}', k) - k + 1) : s.substr(k));
    size_t r = at("run_id", false);   // Learner.cpp:238-239: the metrics run continues under its id
    if (r != std::string::npos && r < s.size() && s[r] == '"') runID = s.substr(r + 1, s.find('"', r + 1) - r - 1);
}

void Learner::Save() {
    Impl& m = *impl;
    if (config.checkpointSaveFolder.empty()) RG_ERR_CLOSE("Learner::Save(): checkpointSaveFolder is empty");
    if (m.learnPending) HOST_HIP(hipEventSynchronize(m.evLearnDone));   // collectionDuringLearn: the epochs still running write the parameters saved here
    std::filesystem::path folder = config.checkpointSaveFolder / std::to_string(totalTimesteps);
    std::filesystem::create_directories(folder);
    SaveStats(folder / "RUNNING_STATS.json");
    const std::vector<int32_t> dPol = DimsOf(m.D, config.ppo.policyLayerSizes, m.A), dCri = DimsOf(m.D, config.ppo.criticLayerSizes, 1);
    const int64_t nPol = rlgpu_learner_num_params(m.lrn, 0), nCri = rlgpu_learner_num_params(m.lrn, 1);
    std::vector<float> pol(nPol), cri(nCri), am(nPol + nCri), av(nPol + nCri); int64_t sp = 0, sc = 0;
    m.LrnCheck(rlgpu_learner_get_params(m.lrn, 0, pol.data()), "get_params"); m.LrnCheck(rlgpu_learner_get_params(m.lrn, 1, cri.data()), "get_params");
    m.LrnCheck(rlgpu_learner_get_adam_state(m.lrn, am.data(), av.data(), &sp, &sc), "get_adam_state");
    // the reference's own payloads: torch::save(Sequential) and Adam::save archives (PPOLearner.cpp:408-411,466-472)
    LtCheck(rlgpu_lt_write_model((folder / "PPO_POLICY.lt").string().c_str(), dPol.data(), (int)dPol.size() - 1, pol.data()), "failed to save model to", folder / "PPO_POLICY.lt");
    LtCheck(rlgpu_lt_write_model((folder / "PPO_CRITIC.lt").string().c_str(), dCri.data(), (int)dCri.size() - 1, cri.data()), "failed to save model to", folder / "PPO_CRITIC.lt");
    LtCheck(rlgpu_lt_write_adam((folder / "PPO_POLICY_OPTIM.lt").string().c_str(), dPol.data(), (int)dPol.size() - 1, config.ppo.policyLR, am.data(), av.data(), sp),
            "failed to save optimizer to", folder / "PPO_POLICY_OPTIM.lt");
    LtCheck(rlgpu_lt_write_adam((folder / "PPO_CRITIC_OPTIM.lt").string().c_str(), dCri.data(), (int)dCri.size() - 1, config.ppo.criticLR, am.data() + nPol, av.data() + nPol, sc),
            "failed to save optimizer to", folder / "PPO_CRITIC_OPTIM.lt");
    m.tsSinceSave = 0;
    if (config.checkpointsToKeep > 0) {   // prune the lowest-numbered folders (Learner.cpp:256-280)
        std::vector<uint64_t> nums;
        for (auto& e : std::filesystem::directory_iterator(config.checkpointSaveFolder)) {
            std::string n = e.path().filename().string();
            if (e.is_directory() && !n.empty() && std::all_of(n.begin(), n.end(), ::isdigit)) nums.push_back(std::stoull(n));
        }
        std::sort(nums.begin(), nums.end());
        for (size_t i = 0; i + config.checkpointsToKeep < nums.size(); i++) std::filesystem::remove_all(config.checkpointSaveFolder / std::to_string(nums[i]));
    }
    RG_LOG("Learner: saved checkpoint " << folder.string());
}

// Learner.cpp:311-370: older checkpoints, one per timestepsPerVersion going back from the loaded one, become the skill tracker's stored
// versions when they carry a "skill_rating"
void Learner::LoadOldVersions(const std::vector<int32_t>& policyDims) {
    const SkillTrackerConfig& sc = config.skillTrackerConfig;
    RG_LOG("Attempting to load " << sc.maxVersions << " old versions for skill tracker...");
    const int64_t targetInterval = sc.timestepsPerVersion, maxAcceptableOverage = targetInterval;
    int64_t targetTimesteps = (int64_t)totalTimesteps;
    for (int i = 0; i < sc.maxVersions; i++) {
        targetTimesteps -= targetInterval;
        std::string bestRating; int64_t bestTimesteps = -1;
        for (auto& entry : std::filesystem::directory_iterator(config.checkpointLoadFolder)) {
            const std::string name = entry.path().filename().string();
            if (!entry.is_directory() || name.empty() || !std::all_of(name.begin(), name.end(), ::isdigit)) continue;
            const int64_t nameVal = std::stoll(name);
            if (nameVal >= targetTimesteps + targetInterval) continue;
            if (bestTimesteps != -1 && std::llabs(nameVal - targetTimesteps) >= std::llabs(bestTimesteps - targetTimesteps)) continue;
            std::ifstream f(entry.path() / "RUNNING_STATS.json");
            if (!f.good()) continue;
            std::string js((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            size_t k = js.find("\"skill_rating\"");
            if (k == std::string::npos) continue;
            k = js.find(':', k) + 1;
            while (k < js.size() && isspace((unsigned char)js[k])) k++;
            bestRating = js[k] == '{' ? js.substr(k, js.find('}', k) - k + 1) : js.substr(k);
            bestTimesteps = nameVal;
        }
        if (bestTimesteps != -1 && bestTimesteps >= targetTimesteps - maxAcceptableOverage) {
            RG_LOG(" > [" << i << "]: Found at " << bestTimesteps << " (target = " << targetTimesteps << ", delta = " << (targetTimesteps - bestTimesteps) << ")");
            std::filesystem::path file = config.checkpointLoadFolder / std::to_string(bestTimesteps) / "PPO_POLICY.lt";
            std::vector<float> params((size_t)rlgpu_learner_num_params(impl->lrn, 0));
            if (std::filesystem::exists(file) && rlgpu_lt_read_model(file.string().c_str(), policyDims.data(), (int)policyDims.size() - 1, params.data()) == RLGPU_OK)
                skillTracker->AppendOldPolicy(params, skillTracker->LoadRatingSet(bestRating));
            else RG_LOG(" > FAILED to load policy, policy does not exist in checkpoint!");
        } else {
            RG_LOG(" > [" << i << "]: None found");
        }
    }
}

void Learner::Load() {
    Impl& m = *impl;
    if (config.checkpointLoadFolder.empty() || !std::filesystem::is_directory(config.checkpointLoadFolder)) return;
    bool any = false; uint64_t best = 0;   // the highest-numbered sub-folder (Learner.cpp:291-309)
    for (auto& e : std::filesystem::directory_iterator(config.checkpointLoadFolder)) {
        std::string n = e.path().filename().string();
        if (e.is_directory() && !n.empty() && std::all_of(n.begin(), n.end(), ::isdigit)) { best = std::max<uint64_t>(best, std::stoull(n)); any = true; }
    }
    if (!any) return;
    std::filesystem::path folder = config.checkpointLoadFolder / std::to_string(best);
    LoadStats(folder / "RUNNING_STATS.json");
    const std::vector<int32_t> dPol = DimsOf(m.D, config.ppo.policyLayerSizes, m.A), dCri = DimsOf(m.D, config.ppo.criticLayerSizes, 1);
    const int64_t nPol = rlgpu_learner_num_params(m.lrn, 0), nCri = rlgpu_learner_num_params(m.lrn, 1);
    if (!std::filesystem::exists(folder / "PPO_POLICY.lt")) RG_ERR_CLOSE("PPOLearner: Failed to find file \"PPO_POLICY.lt\" in " << folder.string() << ".");   // PPOLearner.cpp:414-417
    std::vector<float> pol(nPol), cri(nCri);
    LtCheck(rlgpu_lt_read_model((folder / "PPO_POLICY.lt").string().c_str(), dPol.data(), (int)dPol.size() - 1, pol.data()), "Failed to load model from", folder / "PPO_POLICY.lt");
    m.LrnCheck(rlgpu_learner_set_params(m.lrn, 0, pol.data()), "set_params");
    if (std::filesystem::exists(folder / "PPO_CRITIC.lt")) {   // the critic file is optional (PPOLearner.cpp:421-422)
        LtCheck(rlgpu_lt_read_model((folder / "PPO_CRITIC.lt").string().c_str(), dCri.data(), (int)dCri.size() - 1, cri.data()), "Failed to load model from", folder / "PPO_CRITIC.lt");
        m.LrnCheck(rlgpu_learner_set_params(m.lrn, 1, cri.data()), "set_params");
    }
    std::vector<float> am(nPol + nCri, 0.f), av(nPol + nCri, 0.f); int64_t sp = 0, sc = 0;
    auto readOptim = [&](const char* name, const std::vector<int32_t>& d, float* mm, float* vv, int64_t& step) {
        std::filesystem::path p = folder / name;
        if (!std::filesystem::exists(p)) { RG_LOG("WARNING: No optimizer found at " << p.string() << ", optimizer will be reset"); return; }   // PPOLearner.cpp:436-441
        if (std::filesystem::file_size(p) == 0) { RG_LOG("WARNING: Saved optimizer is empty, optimizer will be reset"); return; }               // :443-449
        LtCheck(rlgpu_lt_read_adam(p.string().c_str(), d.data(), (int)d.size() - 1, mm, vv, &step), "Failed to load optimizers from", p);       // :460-465
    };
    readOptim("PPO_POLICY_OPTIM.lt", dPol, am.data(), av.data(), sp);
    readOptim("PPO_CRITIC_OPTIM.lt", dCri, am.data() + nPol, av.data() + nPol, sc);
    m.LrnCheck(rlgpu_learner_set_adam_state(m.lrn, am.data(), av.data(), sp, sc), "set_adam_state");
    UpdateLearningRates(config.ppo.policyLR, config.ppo.criticLR);   // Learner.cpp:501
    if (skillTracker && config.skillTrackerConfig.loadOldVersionsFromCheckpoints) LoadOldVersions(dPol);
    RG_LOG("Learner: loaded checkpoint " << folder.string() << " (" << totalTimesteps << " timesteps)");
}

}  // namespace RLGPC
