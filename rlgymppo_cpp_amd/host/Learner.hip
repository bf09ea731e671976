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
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <future>
#include <numeric>
#include <thread>

#include <RLGymPPO_CPP/Learner.h>
#include "../../include/rlgpu_state.h"

namespace {

#define HOST_HIP(call)                                                                                   \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess) RG_ERR_CLOSE(#call << " failed: " << hipGetErrorString(_e));               \
    } while (0)

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

std::filesystem::path g_mesh_folder;

template <class T>
T* dev_alloc(size_t n) { T* p = nullptr; HOST_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T))); return p; }

}  // namespace

namespace RocketSim {
void Init(const std::filesystem::path& collisionMeshesFolder, bool silent) {
    g_mesh_folder = collisionMeshesFolder;
    if (!silent) RG_LOG("RocketSim::Init: arena meshes from \"" << (collisionMeshesFolder / "soccar").string() << "\" (procedural soccar mesh if absent)");
}
const std::filesystem::path& GetCollisionMeshFolder() { return g_mesh_folder; }
}

namespace RLGSC {
// GameState::UpdateFromArena / PlayerData::UpdateFromCar (SIM/Utils/Gamestates/GameState.cpp:52-104, PlayerData.cpp:4-34) from a
// downloaded env
GameState::GameState(const RlgpuArenaState& s, int tickSkip) {
    auto V = [](const float* p) { return Vec(p[0], p[1], p[2]); };
    scoreLine.teamGoals[0] = s.gym.score_line[0]; scoreLine.teamGoals[1] = s.gym.score_line[1];
    lastTouchCarID = s.gym.last_touch_car_id;
    lastTickCount = (uint64_t)s.tick_count; deltaTickCount = tickSkip;
    ball.pos = V(s.ball.pos); ball.vel = V(s.ball.vel); ball.angVel = V(s.ball.ang_vel);
    ballInv = ball.Invert();
    // RLGym pad order -> RocketSim pad index: the map GameState.cpp:10-50 builds by matching CommonValues::BOOST_LOCATIONS against
    // the arena's pads (a constant of the two tables; the device obs builder uses the same one)
    static const int8_t PAD_ORDER[RLGPU_NUM_PADS] = {6, 7, 8, 4, 5, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 0, 19, 20, 1, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 2, 3, 31, 32, 33};
    for (int p = 0; p < RLGPU_NUM_PADS; p++) { boostPads[p] = s.pads[PAD_ORDER[p]].is_active != 0; boostPadsInv[RLGPU_NUM_PADS - 1 - p] = boostPads[p]; }
    players.resize(s.num_cars);
    for (int k = 0; k < s.num_cars; k++) {
        const RlgpuCarState& c = s.cars[k]; const RlgpuPlayerGymState& g = s.gym.players[k];
        PlayerData& pd = players[k];
        pd.carId = (uint32_t)(k + 1); pd.team = (k % 2 == 0) ? Team::BLUE : Team::ORANGE;
        pd.phys.pos = V(c.pos); pd.phys.vel = V(c.vel); pd.phys.angVel = V(c.ang_vel);
        pd.phys.rotMat.forward = V(c.rot); pd.phys.rotMat.right = V(c.rot + 3); pd.phys.rotMat.up = V(c.rot + 6);
        pd.physInv = pd.phys.Invert();
        CarState& cs = pd.carState;
        cs.pos = pd.phys.pos; cs.vel = pd.phys.vel; cs.angVel = pd.phys.angVel; cs.rotMat = pd.phys.rotMat;
        cs.isOnGround = c.flags & RLGPU_CF_ON_GROUND; cs.hasJumped = c.flags & RLGPU_CF_HAS_JUMPED; cs.hasDoubleJumped = c.flags & RLGPU_CF_HAS_DOUBLE_JUMPED;
        cs.hasFlipped = c.flags & RLGPU_CF_HAS_FLIPPED; cs.isJumping = c.flags & RLGPU_CF_IS_JUMPING; cs.isFlipping = c.flags & RLGPU_CF_IS_FLIPPING;
        cs.isSupersonic = c.flags & RLGPU_CF_IS_SUPERSONIC; cs.isDemoed = c.flags & RLGPU_CF_IS_DEMOED;
        cs.boost = c.boost; cs.airTimeSinceJump = c.air_time_since_jump; cs.jumpTime = c.jump_time; cs.flipTime = c.flip_time; cs.demoRespawnTimer = c.demo_respawn_timer;
        pd.matchGoals = g.match_goals; pd.matchSaves = g.match_saves; pd.matchAssists = g.match_assists; pd.matchShots = g.match_shots;
        pd.matchShotPasses = g.match_shot_passes; pd.matchBumps = g.match_bumps; pd.matchDemos = g.match_demos; pd.boostPickups = g.boost_pickups;
        pd.boostFraction = c.boost / 100.f;
        // PlayerData.cpp:20-30
        pd.ballTouchedStep = (c.flags & RLGPU_CF_BALLHIT_VALID) && c.bh_tick_hit >= s.tick_count - tickSkip;
        pd.ballTouchedTick = (c.flags & RLGPU_CF_BALLHIT_VALID) && c.bh_tick_hit == s.tick_count - 1;
        pd.hasJump = !cs.hasJumped;
        pd.hasFlip = !cs.hasDoubleJumped && !cs.hasFlipped && cs.airTimeSinceJump < 1.25f;   // RLConst::DOUBLEJUMP_MAX_DELAY
    }
}
}  // namespace RLGSC

namespace RLGPC {

struct Learner::Impl {
    rlgpu_env* env = nullptr; rlgpu_learner* lrn = nullptr; rlgpu_shuffler* shuf = nullptr;
    rlgpu_comm* comm = nullptr; int rank = 0, world = 1, device = 0;
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
    Timer renderTimer;
    uint64_t cumulativeModelUpdates = 0, tsSinceSave = 0;
    std::vector<GameInst> games;
    std::vector<RlgpuArenaState> hostStates; std::vector<float> hostRew; std::vector<int32_t> hostDone;
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
            const char* lr = std::getenv("LOCAL_RANK");
            m.device = lr ? std::atoi(lr) : m.rank;
        }
        HOST_HIP(hipSetDevice(m.device));
    }
    EnvCreateResult ecr = envCreateFn();
    if (!ecr.match || !ecr.gym) RG_ERR_CLOSE("EnvCreateFn returned a null match or gym");
    m.match = ecr.match; m.gym = ecr.gym; m.tickSkip = ecr.gym->tickSkip;
    RlgpuGymConfig gcfg = m.match->ToDeviceConfig(m.tickSkip);
    gcfg.seed_lo = (uint32_t)config.randomSeed + 1000u * (uint32_t)m.rank; gcfg.seed_hi = 0;   // every rank its own env RNG streams; rank 0 = the single-GPU run
    m.nEnvs = config.numThreads * config.numGamesPerThread;
    m.nPlayers = m.match->playerAmount;
    int rc = rlgpu_env_create(&m.env, m.device, m.nEnvs, m.match->teamSize, &gcfg);
    m.EnvCheck(rc, "create");
    std::filesystem::path soccar = RocketSim::GetCollisionMeshFolder() / "soccar";
    if (!RocketSim::GetCollisionMeshFolder().empty() && std::filesystem::is_directory(soccar)) m.EnvCheck(rlgpu_env_load_cmf_dir(m.env, soccar.string().c_str()), "load_cmf_dir");
    else {
        RG_LOG("Learner: no collision meshes at \"" << soccar.string() << "\" -- using the procedural soccar mesh");
        m.EnvCheck(rlgpu_env_set_procedural_mesh(m.env), "set_procedural_mesh");
    }
    m.nAgents = rlgpu_env_num_agents(m.env); m.D = rlgpu_env_obs_size(m.env); m.A = rlgpu_env_num_actions(m.env);
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
    lc.temperature = config.ppo.policyTemperature; lc.use_bf16 = config.ppo.autocastLearn ? 1 : 0;
    lc.seed_lo = (uint32_t)config.randomSeed; lc.seed_hi = 0; lc.max_rows = m.maxRows;
    rc = rlgpu_learner_create(&m.lrn, m.device, &lc);
    m.LrnCheck(rc, "learner_create");
    // identical parameters on every rank (same init seed), independent exploration: the action sampler is keyed on the rank
    m.LrnCheck(rlgpu_learner_set_sampler(m.lrn, (uint32_t)m.rank, 0), "learner_set_sampler");
    m.retShare = dev_alloc<float>((size_t)std::max(1, config.maxReturnsPerStatsInc));
    rlgpu_shuffler_create(&m.shuf, (uint32_t)config.randomSeed);

    const size_t TN = (size_t)m.T * m.nAgents;
    m.obs = dev_alloc<float>((size_t)(m.T + 1) * m.nAgents * m.D);
    m.acts = dev_alloc<int32_t>(TN); m.done = dev_alloc<int32_t>(TN);
    if (rlgpu_expbuf_create(&m.fifo, config.expBufferSize, m.T, m.nAgents) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: bad expBufferSize " << config.expBufferSize);
    const size_t EX = (size_t)rlgpu_expbuf_num_slots(m.fifo) * TN;
    m.idx = dev_alloc<int32_t>(EX);
    m.exObs = dev_alloc<float>(EX * m.D); m.exActs = dev_alloc<int32_t>(EX); m.exLogp = dev_alloc<float>(EX); m.exAdv = dev_alloc<float>(EX); m.exTgt = dev_alloc<float>(EX);
    m.logp = dev_alloc<float>(TN); m.rew = dev_alloc<float>(TN); m.doneF = dev_alloc<float>(TN); m.trunc = dev_alloc<float>(TN);
    m.adv = dev_alloc<float>(TN); m.tgt = dev_alloc<float>(TN); m.ret = dev_alloc<float>(TN);
    m.vals = dev_alloc<float>(TN + m.nAgents); m.metrics = dev_alloc<float>(8); m.scratch = dev_alloc<float>(8);
    m.phys[0].resize(EX); m.phys[1].resize(EX);
    m.EnvCheck(rlgpu_env_reset(m.env, 1, m.ObsAt(0)), "reset");

    if (config.saveFolderAddUnixTimestamp && !config.checkpointSaveFolder.empty())
        config.checkpointSaveFolder += "-" + std::to_string(std::chrono::duration_cast<std::chrono::seconds>(std::chrono::system_clock::now().time_since_epoch()).count());
    if (config.renderMode) renderSender = new RenderSender();                                                                       // Learner.cpp:128-135
    if (config.skillTrackerConfig.enabled) {                                                                                          // Learner.cpp:136-143
        if (config.skillTrackerConfig.envCreateFunc == NULL) config.skillTrackerConfig.envCreateFunc = envCreateFn;
        skillTracker = new SkillTracker(config.skillTrackerConfig, m.lrn, m.D, m.A, config.ppo.policyLayerSizes, config.randomSeed, renderSender);
    }
    if (!config.checkpointLoadFolder.empty()) Load();
    if (config.sendMetrics && m.rank == 0) {                                                                                          // Learner.cpp:149-155
        if (!runID.empty()) RG_LOG("\tRun ID: " << runID);
        metricSender = new MetricSender(config.metricsProjectName, config.metricsGroupName, config.metricsRunName, runID);
    }
}

Learner::~Learner() {
    Impl& m = *impl;
    for (void* p : {(void*)m.obs, (void*)m.acts, (void*)m.done, (void*)m.idx, (void*)m.logp, (void*)m.rew, (void*)m.doneF, (void*)m.trunc, (void*)m.adv, (void*)m.tgt,
                    (void*)m.ret, (void*)m.vals, (void*)m.metrics, (void*)m.scratch, (void*)m.exObs, (void*)m.exActs, (void*)m.exLogp, (void*)m.exAdv, (void*)m.exTgt})
        if (p) (void)hipFree(p);
    if (m.drawPending) m.nextDraw.wait();
    delete skillTracker; delete metricSender; delete renderSender;
    if (m.fifo) rlgpu_expbuf_destroy(m.fifo);
    if (m.shuf) rlgpu_shuffler_destroy(m.shuf);
    if (m.lrn) rlgpu_learner_destroy(m.lrn);
    if (m.env) rlgpu_env_destroy(m.env);
    if (m.retShare) (void)hipFree(m.retShare);
    if (m.comm) rlgpu_comm_destroy(m.comm);
    delete m.gym; delete m.match;   // GameInst deletes its gym and match in the reference (GameInst.h:53-56); plugins stay the user's
    delete impl;
}

int Learner::NumEnvs() const { return impl->nEnvs; }
int Learner::NumAgents() const { return impl->nAgents; }
int Learner::Rank() const { return impl->rank; }
int Learner::WorldSize() const { return impl->world; }
double Learner::MaxOverRanks(double v) {
    Impl& m = *impl;
    if (!m.comm) return v;
    std::vector<float> h(m.world, 0.f); h[m.rank] = (float)v;
    float* d = dev_alloc<float>(m.world);
    HOST_HIP(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    if (rlgpu_comm_allreduce_f32(m.comm, d, m.world, nullptr) != RLGPU_OK) RG_ERR_CLOSE("rlgpu_comm_allreduce_f32: " << rlgpu_comm_last_error(m.comm));
    HOST_HIP(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return (double)*std::max_element(h.begin(), h.end());
}
bool Learner::UsesFusedCollection() const { return impl->fusedCollect && impl->match->teamSize <= 2 && !stepCallback && !renderSender; }
void Learner::DeviceTimings(float& envMs, int& envLaunches, float& gemmMs, double& gemmFlops, int& gemmCalls, bool reset) {
    impl->EnvCheck(rlgpu_env_timing_total(impl->env, &envMs, &envLaunches, reset ? 1 : 0), "timing_total");
    impl->LrnCheck(rlgpu_learner_timing_total(impl->lrn, &gemmMs, &gemmFlops, &gemmCalls, reset ? 1 : 0), "learner_timing_total");
}

void Learner::UpdateLearningRates(float policyLR, float criticLR) {
    config.ppo.policyLR = policyLR; config.ppo.criticLR = criticLR;
    impl->LrnCheck(rlgpu_learner_set_lr(impl->lrn, policyLR, criticLR), "learner_set_lr");
}

// ThreadAgent::_RunFunc (ThreadAgent.cpp:24-195) for every game at once: policy inference and the env step never leave the device
void Learner::CollectTimesteps() {
    Impl& m = *impl;
    const size_t rowObs = (size_t)m.nAgents * m.D;
    if (!m.first) HOST_HIP(hipMemcpyAsync(m.ObsAt(0), m.ObsAt(m.T), rowObs * 4, hipMemcpyDeviceToDevice, nullptr));
    m.first = false;
    const bool slow = (bool)stepCallback;
    if (slow && m.games.empty()) {
        m.games.resize(m.nEnvs);
        for (int e = 0; e < m.nEnvs; e++) { m.games[e].gym = m.gym; m.games[e].match = m.match; m.games[e].index = e; }
        m.hostStates.resize(m.nEnvs); m.hostRew.resize(m.nAgents); m.hostDone.resize(m.nAgents);
    }
    // no per-step host work: the whole phase in one launch (rlgpu_collect), when the policy fits the in-kernel inference
    if (!slow && !renderSender && m.fusedCollect && m.match->teamSize <= 2) {   // 3v3: one env per wavefront, the in-kernel inference does not amortise
        int rc = rlgpu_collect(m.env, m.lrn, m.T, m.obs, m.acts, m.logp, m.rew, m.done, config.deterministic ? 1 : 0);
        if (rc == RLGPU_OK) { totalTimesteps += (uint64_t)m.B * (uint64_t)m.world; return; }
        if (rc != RLGPU_ERR_STATE) m.EnvCheck(rc, "collect");
        m.fusedCollect = false;   // fp32 mode or a net too wide for the kernel's LDS scratch: alternate act / step
    }
    for (int t = 0; t < m.T; t++) {
        const size_t o = (size_t)t * m.nAgents;
        m.LrnCheck(rlgpu_policy_act(m.lrn, m.ObsAt(t), m.nAgents, config.deterministic ? 1 : 0, nullptr, m.acts + o, m.logp + o), "policy_act");
        m.EnvCheck(rlgpu_env_step(m.env, m.acts + o, m.ObsAt(t + 1), m.rew + o, m.done + o), "step");
        if (renderSender) RenderStep(t);
        if (slow) {
            // the reference hands every game's StepResult to the callback (GameInst.cpp:14-38): materialise host GameStates -- slow path.
            // NB: the downloaded state is the one AFTER the step's auto-reset when the episode ended.
            m.EnvCheck(rlgpu_env_download_states(m.env, m.hostStates.data(), nullptr, m.nEnvs), "download_states");
            HOST_HIP(hipMemcpy(m.hostRew.data(), m.rew + o, (size_t)m.nAgents * 4, hipMemcpyDeviceToHost));
            HOST_HIP(hipMemcpy(m.hostDone.data(), m.done + o, (size_t)m.nAgents * 4, hipMemcpyDeviceToHost));
            for (int e = 0; e < m.nEnvs; e++) {
                GameInst& g = m.games[e];
                RLGSC::Gym::StepResult sr;
                sr.state = RLGSC::GameState(m.hostStates[e], m.tickSkip);
                sr.reward.assign(m.hostRew.begin() + (size_t)e * m.nPlayers, m.hostRew.begin() + (size_t)(e + 1) * m.nPlayers);
                sr.done = m.hostDone[(size_t)e * m.nPlayers] != 0;
                float sum = std::accumulate(sr.reward.begin(), sr.reward.end(), 0.f);
                g.avgStepRew.Add(sum, (uint64_t)m.nPlayers); g.curEpRew += sum / m.nPlayers; g.totalSteps++;
                if (sr.done) { g.avgEpRew += g.curEpRew; g.curEpRew = 0; }
                stepCallback(&g, sr, g._metrics);
            }
        }
    }
    totalTimesteps += (uint64_t)m.B * (uint64_t)m.world;
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

void Learner::AddNewExperience(Report& report) {
    Impl& m = *impl;
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
    HOST_HIP(hipMemsetAsync(m.metrics, 0, 32, nullptr));
    int nMini = 0, nUpdates = 0;
    Timer t;
    for (int ep = 0; ep < config.ppo.epochs; ep++) {
        // ExperienceBuffer::GetAllBatchesShuffled (ExperienceBuffer.cpp:104-126) over the whole FIFO: logical rows are oldest iteration
        // first and agent-major inside one (trajectory after trajectory); the device slots are time-major
        const int64_t cur = m.TakeDraw();
        HOST_HIP(hipMemcpyAsync(m.idx, m.phys[m.physFlip].data(), (size_t)cur * 4, hipMemcpyHostToDevice, nullptr));   // pageable source: staged before the call returns
        m.physFlip ^= 1;
        if (ep == config.ppo.epochs - 1 && rlgpu_expbuf_submit(m.fifo, &m.pendingSlot) != RLGPU_OK) RG_ERR_CLOSE("ExperienceBuffer: no free slot");   // the next draw sees the FIFO after the next submit
        m.StartDraw();
        for (int64_t b = 0; b + m.batch <= cur; b += m.batch) {   // the remainder is dropped (ExperienceBuffer.cpp:115-117)
            m.LrnCheck(rlgpu_zero_grads(m.lrn), "zero_grads");
            for (int64_t k = 0; k < m.batch; k += m.mini) {
                m.LrnCheck(rlgpu_ppo_minibatch(m.lrn, m.exObs, m.exActs, m.exLogp, m.exAdv, m.exTgt, m.idx + b + k, (int)m.mini, (float)m.mini / (float)m.batch, m.metrics), "ppo_minibatch");
                nMini++;
            }
            // multi-GPU: ONE all-reduce(sum) of the flat [policy | critic] gradient on the learner's stream, then scale by 1 / world INSIDE
            // the clip so the norm is taken of the global-batch gradient, like a single learner on the union would
            if (m.comm) m.LrnCheck(rlgpu_allreduce_grads(m.lrn, m.comm), "allreduce_grads");
            m.LrnCheck(rlgpu_clip_adam_step(m.lrn, 0.5f, 1.f / (float)m.world), "clip_adam_step");   // clip_grad_norm_(0.5) per network, then Adam (PPOLearner.cpp:273-288)
            nUpdates++;
        }
    }
    m.LrnCheck(rlgpu_learner_sync(m.lrn), "learner_sync");
    float h[8];
    HOST_HIP(hipMemcpy(h, m.metrics, 32, hipMemcpyDeviceToHost));
    const double rows = std::max<double>(1, (double)nMini * m.mini);
    report["Policy Entropy"] = h[0] / rows; report["Mean KL Divergence"] = h[1] / rows; report["SB3 Clip Fraction"] = h[2] / rows;
    report["Value Function Loss"] = h[4] / rows; report["PPO Learn Time"] = t.Elapsed();
    totalEpochs += config.ppo.epochs; m.cumulativeModelUpdates += nUpdates;
    report["Cumulative Model Updates"] = (double)m.cumulativeModelUpdates;
}

void Learner::Learn() {
    Impl& m = *impl;
    RG_LOG("Learner: " << m.nEnvs << " envs (" << m.nAgents << " agents), obs " << m.D << ", actions " << m.A << ", " << m.T << " steps/env/iteration = "
           << m.B << " timesteps, batch " << m.batch << ", minibatch " << m.mini);
    while (config.timestepLimit == 0 || totalTimesteps < config.timestepLimit) {
        if (m.renderOnly) { CollectTimesteps(); continue; }   // render mode: play the (loaded) policy forever, one step per "iteration"
        Report report;
        Timer tAll, tCollect;
        CollectTimesteps();
        HOST_HIP(hipDeviceSynchronize());
        double collectTime = tCollect.Elapsed();
        Timer tConsume;
        AddNewExperience(report);
        LearnPPO(report);
        double consumeTime = tConsume.Elapsed();
        if (skillTracker) {   // Learner.cpp:527-538
            RG_LOG("Running skill eval game(s)...");
            if (config.skillTrackerConfig.stepCallback == NULL) skillTracker->config.stepCallback = stepCallback;
            skillTracker->RunGames((int64_t)m.B);
            for (auto& pair : skillTracker->curRating.data) report[std::string("Skill Rating") + (pair.first.empty() ? "" : " ") + pair.first] = pair.second;
        }
        totalIterations++;
        report["Total Iterations"] = (double)totalIterations; report["Cumulative Timesteps"] = (double)totalTimesteps;
        report["Timesteps Collected"] = (double)m.B;
        report["Collection Time"] = collectTime; report["Consumption Time"] = consumeTime; report["Total Iteration Time"] = tAll.Elapsed();
        report["Collected Steps/Second"] = (double)m.B * m.world / std::max(collectTime, 1e-9);
        report["Overall Steps/Second"] = (double)m.B * m.world / std::max(tAll.Elapsed(), 1e-9);
        if (iterationCallback) iterationCallback(this, report);
        if (config.sendMetrics && metricSender) metricSender->Send(report);                                                                       // Learner.cpp:589-590
        if (m.rank != 0 || std::getenv("RLGPU_QUIET")) { m.tsSinceSave += (uint64_t)m.B * (uint64_t)m.world; if (m.rank == 0 && !config.checkpointSaveFolder.empty() && m.tsSinceSave >= (uint64_t)std::max<int64_t>(config.timestepsPerSave, 1)) Save(); continue; }
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
        m.tsSinceSave += (uint64_t)m.B * (uint64_t)m.world;
        if (m.rank == 0 && !config.checkpointSaveFolder.empty() && m.tsSinceSave >= (uint64_t)std::max<int64_t>(config.timestepsPerSave, 1)) Save();
    }
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
      << ",\n    \"epoch\": " << totalEpochs << ",\n    \"reward_running_stats\": {\n        \"count\": " << returnStats.count << ",\n        \"mean\": [\n            "
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
    size_t k = at("skill_rating", false);   // Learner.cpp:229-231
    if (skillTracker && k != std::string::npos) skillTracker->curRating = skillTracker->LoadRatingSet(s[k] == '{' ? s.substr(k, s.find('}', k) - k + 1) : s.substr(k));
    size_t r = at("run_id", false);   // Learner.cpp:238-239: the metrics run continues under its id
    if (r != std::string::npos && r < s.size() && s[r] == '"') runID = s.substr(r + 1, s.find('"', r + 1) - r - 1);
}

void Learner::Save() {
    Impl& m = *impl;
    if (config.checkpointSaveFolder.empty()) RG_ERR_CLOSE("Learner::Save(): checkpointSaveFolder is empty");
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
