// The host path of an env batch: plugin kinds without a device form (Match::DevicePlan -- a user's reward function, terminal condition, obs
// builder, state setter or action parser) and step callbacks run on the host between the device's steps, everything else stays in the step
// kernel.  Shared by the training batch (Learner.hip) and the skill tracker's eval batch (SkillTracker.hip); not part of the public headers.
//
// Follows Gym::Step (SIM/Gym.cpp:68-102) and GameInst::Step (PUB/Threading/GameInst.cpp:7-38) per env, with the arena work of all envs done
// by one launch: the device steps (rlgpu_env_step / _step_controls), every env's arena where Gym::Step builds its GameState comes back as a
// snapshot, the host plugins see that GameState, and the envs whose episode ended are reset (state setter on the host facade or on the device).
#pragma once
#include "host_util.h"
#include <algorithm>
#include <array>
#include <exception>
#include <functional>
#include <thread>
#include <vector>

namespace RLGPC {

struct HostEnvPath {
    rlgpu_env* env = nullptr;
    RLGSC::Match::DevicePlan plan;
    int nEnvs = 0, nPlayers = 0, nAgents = 0, tickSkip = 8;
    int D = 0, Ddev = 0;      // observation width the policy sees / the device builder's (they differ only with a host obs builder)
    int workers = 1;
    bool ready = false;
    std::vector<RLGSC::Match*> envMatch; std::vector<RLGSC::Gym*> envGym;   // one plugin set per env, like GameInst's; [0] = the caller's match / gym
    float *devObs = nullptr, *devControls = nullptr;   // the device builder's rows when the obs builder runs on the host; host-parsed controls
    std::vector<RlgpuArenaState> snaps, fresh;
    std::vector<RLGSC::GameState> prevGs;
    std::vector<float> hObs, hRew, hControls; std::vector<int32_t> hDone, hActs;
    std::vector<RLGSC::Arena*> arenas;                 // scratch facades for user state setters, one per worker
    std::function<float*(size_t)> allocF32;            // device allocator of the owner (the learner's is redzone-aware); hipMalloc when empty
    bool deviceResets = false;                         // the step kernel resets an env whose episode ended itself (RlgpuGymConfig::host_resets = 0 although a plugin
                                                       // runs on the host: the deferred-reward mode of the Learner when it has to step by step after all)

    HostEnvPath() = default;
    HostEnvPath(const HostEnvPath&) = delete;
    HostEnvPath& operator=(const HostEnvPath&) = delete;
    ~HostEnvPath() {
        for (size_t e = 1; e < envMatch.size(); e++) if (envMatch[e] != envMatch[0]) { delete envGym[e]; delete envMatch[e]; }   // per-env plugin sets (not the aliases of a callback-only run)
        for (RLGSC::Arena* a : arenas) delete a;
        if (devObs) (void)hipFree(devObs);
        if (devControls) (void)hipFree(devControls);
        if (recRing) (void)hipHostFree(recRing);
    }

    void EnvCheck(int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("rlgpu_env_" << what << " failed (" << rc << "): " << rlgpu_env_last_error(env)); }
    float* Alloc(size_t n) {
        if (allocF32) return allocF32(n);
        float* p = nullptr; HOST_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(float))); return p;
    }

    // fn(env, index in ids, worker) for every env in `ids` (all envs when null) on `workers` host threads; the first exception is rethrown here
    template <class F>
    void ForEnvs(const std::vector<int32_t>* ids, F fn) {
        const int n = ids ? (int)ids->size() : nEnvs;
        const int nw = std::max(1, std::min(workers, n));
        std::vector<std::exception_ptr> errs(nw);
        auto run = [&](int w) {
            try { for (int i = w; i < n; i += nw) fn(ids ? (*ids)[i] : i, i, w); } catch (...) { errs[w] = std::current_exception(); }
        };
        std::vector<std::thread> th;
        for (int w = 1; w < nw; w++) th.emplace_back(run, w);
        run(0);
        for (auto& t : th) t.join();
        for (auto& e : errs) if (e) std::rethrow_exception(e);
    }

    // `match` / `gym`: the set the caller made with `create` already (env 0's).  With host plugins every further env gets a set of its own --
    // user plugins may carry per-episode state, as every GameInst's do (Learner.cpp:99-109).
    void Setup(rlgpu_env* env_, const RLGSC::Match::DevicePlan& plan_, RLGSC::Match* match, RLGSC::Gym* gym, const EnvCreateFn& create, int numThreads,
               int nEnvs_, int nPlayers_, int nAgents_, int D_, int Ddev_, int tickSkip_) {
        if (ready) return;
        ready = true;
        env = env_; plan = plan_; nEnvs = nEnvs_; nPlayers = nPlayers_; nAgents = nAgents_; D = D_; Ddev = Ddev_; tickSkip = tickSkip_;
        workers = std::max(1, std::min({numThreads, (int)std::thread::hardware_concurrency(), 64}));
        EnvCheck(rlgpu_env_enable_snapshots(env, 1), "enable_snapshots");
        snaps.resize(nEnvs); prevGs.resize(nEnvs); hRew.resize(nAgents); hDone.resize(nAgents); hActs.resize(nAgents);
        envMatch.assign(nEnvs, match); envGym.assign(nEnvs, gym);
        if (plan.AnyHost()) {
            for (int e = 1; e < nEnvs; e++) {
                EnvCreateResult r = create();
                if (!r.match || !r.gym) RG_ERR_CLOSE("EnvCreateFn returned a null match or gym");
                envMatch[e] = r.match; envGym[e] = r.gym;
            }
            for (int w = 0; w < workers; w++) arenas.push_back(RLGSC::MakeScratchArena(match->teamSize, match->spawnOpponents));
            if (plan.hostObs) devObs = Alloc((size_t)nAgents * Ddev);
            if (plan.hostParser) { devControls = Alloc((size_t)nAgents * 8); hControls.resize((size_t)nAgents * 8); }
        }
    }

    // GameInst::Step's `gym->Reset()` for the listed envs (GameInst.cpp:27-32, Gym.cpp:58-66): state setter (host or device), the device's
    // episode bookkeeping, then the host plugins' Reset hooks and -- with a host obs builder -- the first observation rows
    void ResetEnvs(const std::vector<int32_t>& ids, float* obsRows, bool deviceDidReset) {
        const int n = (int)ids.size();
        if (n == 0) return;
        float* devRows = plan.hostObs ? devObs : obsRows;
        fresh.resize(n);
        // the pads as the state setter's GameState showed them: Match::ResetState resets them only after the setter returned (Match.cpp:55-69), so the
        // new episode's first GameState / observation still carries the previous episode's pad states
        std::vector<std::array<uint8_t, RLGPU_NUM_PADS>> padsBefore;
        if (plan.AnyHost() && !deviceDidReset) {   // (the envs are still as their episodes ended: RlgpuGymConfig::host_resets)
            EnvCheck(rlgpu_env_download_states(env, fresh.data(), ids.data(), n), "download_states");
            padsBefore.resize(n);
            for (int i = 0; i < n; i++) for (int p = 0; p < RLGPU_NUM_PADS; p++) padsBefore[i][p] = fresh[i].pads[p].is_active;
        }
        if (plan.hostSetter) {
            if (padsBefore.empty()) EnvCheck(rlgpu_env_download_states(env, fresh.data(), ids.data(), n), "download_states");
            ForEnvs(&ids, [&](int e, int i, int w) {
                RLGSC::Arena* arena = arenas[w];
                arena->_state = fresh[i]; arena->_SyncFromState();
                const RLGSC::GameState gs = envMatch[e]->stateSetter->ResetState(arena);   // (the pad reset of Match::ResetState is the device's: gym_episode_reset)
                if ((int)gs.players.size() != nPlayers) RG_ERR_CLOSE("Match::ResetState(): New state has a different amount of players, expected " << nPlayers << " but got " << gs.players.size() << ".");
                arena->_SyncToState();
                fresh[i] = arena->_state;
                if (!padsBefore.empty()) for (int p = 0; p < RLGPU_NUM_PADS; p++) padsBefore[i][p] = fresh[i].pads[p].is_active;   // (a setter may have touched them)
            });
            EnvCheck(rlgpu_env_upload_states(env, fresh.data(), ids.data(), n), "upload_states");
            EnvCheck(rlgpu_env_reset_envs(env, ids.data(), n, 0, devRows), "reset_envs");
        } else if (!deviceDidReset) {
            EnvCheck(rlgpu_env_reset_envs(env, ids.data(), n, 1, devRows), "reset_envs");
            // the device's setters are the built-in ones, and both reset the pads BEFORE they build the episode's first GameState (KickoffState and
            // RandomState through Arena::ResetToRandomKickoff, RandomState.cpp:11, Arena.cpp:209-210): that GameState shows every pad active
            for (auto& pb : padsBefore) pb.fill(1);
        }
        if (!plan.AnyHost()) return;
        EnvCheck(rlgpu_env_download_states(env, fresh.data(), ids.data(), n), "download_states");
        if (plan.hostObs && hObs.size() < (size_t)nAgents * D) hObs.resize((size_t)nAgents * D);
        ForEnvs(&ids, [&](int e, int i, int) {
            RlgpuArenaState st = fresh[i];
            if (!padsBefore.empty()) for (int p = 0; p < RLGPU_NUM_PADS; p++) st.pads[p].is_active = padsBefore[i][p];
            RLGSC::GameState gs0(st, (int)st.tick_count);
            envMatch[e]->EpisodeReset(gs0);
            prevGs[e] = gs0;
            if (plan.hostObs) {
                const RLGSC::FList2 rows = envMatch[e]->BuildObservations(gs0);
                for (int k = 0; k < nPlayers; k++) {
                    if ((int)rows[k].size() != D) RG_ERR_CLOSE("OBSBuilder::BuildOBS returned " << rows[k].size() << " values, the first observation had " << D);
                    std::copy(rows[k].begin(), rows[k].end(), hObs.begin() + ((size_t)e * nPlayers + k) * D);
                }
            }
        });
        if (plan.hostObs)
            for (int32_t e : ids) HOST_HIP(hipMemcpy(obsRows + (size_t)e * nPlayers * D, hObs.data() + (size_t)e * nPlayers * D, (size_t)nPlayers * D * 4, hipMemcpyHostToDevice));
    }

    // ---- deferred plugins (LearnerConfig::deferHostRewards): the steps of a finished collection launch, replayed per env in order ----------------------
    // A GameState from a step record (rlgpu_state.h RlgpuStepHead / RlgpuStepCar = what GameState::UpdateFromArena / PlayerData::UpdateFromCar read),
    // filled IN PLACE: the object and its players vector are reused from step to step.
    // (deltaTickCount: ticks since the env's previous GameState; a reset state's is its tick count, as GameState(arena) after Gym::Reset has it)
    static void FillGameState(RLGSC::GameState& gs, const uint32_t* rec, uint64_t prevTick, bool resetState) {
        const RlgpuStepHead& h = *reinterpret_cast<const RlgpuStepHead*>(rec);
        if (h.num_cars < 1 || h.num_cars > RLGPU_MAX_CARS) RG_ERR_CLOSE("step record with " << h.num_cars << " cars: not a record the collection kernel wrote");
        const RlgpuStepCar* cars = reinterpret_cast<const RlgpuStepCar*>(rec + RLGPU_STEP_HEAD_WORDS);
        auto V = [](const float* p) { return Vec(p[0], p[1], p[2]); };
        gs.scoreLine.teamGoals[0] = h.score_line[0]; gs.scoreLine.teamGoals[1] = h.score_line[1];
        gs.lastTouchCarID = h.last_touch_car_id;
        gs.lastTickCount = (uint64_t)h.tick_count; gs.deltaTickCount = resetState ? (int)h.tick_count : (int)((uint64_t)h.tick_count - prevTick);
        gs.ball.pos = V(h.ball_pos); gs.ball.vel = V(h.ball_vel); gs.ball.angVel = V(h.ball_ang_vel);
        gs.ball.rotMat.forward = Vec(1, 0, 0); gs.ball.rotMat.right = Vec(0, 1, 0); gs.ball.rotMat.up = Vec(0, 0, 1);   // (the device's setters leave a BallState's default rotMat)
        gs.ballInv = gs.ball.Invert();
        static const int8_t PAD_ORDER[RLGPU_NUM_PADS] = {6, 7, 8, 4, 5, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 0, 19, 20, 1, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 2, 3, 31, 32, 33};
        const uint64_t pm = (uint64_t)h.pads_active[0] | ((uint64_t)h.pads_active[1] << 32);
        for (int p = 0; p < RLGPU_NUM_PADS; p++) { gs.boostPads[p] = (pm >> PAD_ORDER[p]) & 1ull; gs.boostPadsInv[RLGPU_NUM_PADS - 1 - p] = gs.boostPads[p]; }
        gs.players.resize((size_t)h.num_cars);
        for (int k = 0; k < h.num_cars; k++) {
            const RlgpuStepCar& c = cars[k];
            RLGSC::PlayerData& pd = gs.players[(size_t)k];
            pd.carId = (uint32_t)(k + 1); pd.team = (k % 2 == 0) ? Team::BLUE : Team::ORANGE;
            pd.phys.pos = V(c.pos); pd.phys.vel = V(c.vel); pd.phys.angVel = V(c.ang_vel);
            pd.phys.rotMat.forward = V(c.rot); pd.phys.rotMat.right = V(c.rot + 3); pd.phys.rotMat.up = V(c.rot + 6);
            pd.physInv = pd.phys.Invert();
            CarState& cs = pd.carState;
            cs = CarState();
            cs.pos = pd.phys.pos; cs.vel = pd.phys.vel; cs.angVel = pd.phys.angVel; cs.rotMat = pd.phys.rotMat;
            cs.isOnGround = c.flags & RLGPU_CF_ON_GROUND; cs.hasJumped = c.flags & RLGPU_CF_HAS_JUMPED; cs.hasDoubleJumped = c.flags & RLGPU_CF_HAS_DOUBLE_JUMPED;
            cs.hasFlipped = c.flags & RLGPU_CF_HAS_FLIPPED; cs.isJumping = c.flags & RLGPU_CF_IS_JUMPING; cs.isFlipping = c.flags & RLGPU_CF_IS_FLIPPING;
            cs.isSupersonic = c.flags & RLGPU_CF_IS_SUPERSONIC; cs.isDemoed = c.flags & RLGPU_CF_IS_DEMOED;
            cs.boost = c.boost; cs.airTimeSinceJump = c.air_time_since_jump; cs.jumpTime = c.jump_time; cs.flipTime = c.flip_time; cs.demoRespawnTimer = c.demo_respawn_timer;
            pd.matchGoals = c.counters[0]; pd.matchSaves = c.counters[1]; pd.matchAssists = c.counters[2]; pd.matchShots = c.counters[3];
            pd.matchShotPasses = c.counters[4]; pd.matchBumps = c.counters[5]; pd.matchDemos = c.counters[6]; pd.boostPickups = c.counters[7];
            pd.boostFraction = c.boost / 100.f;
            pd.ballTouchedStep = c.touched & 1u; pd.ballTouchedTick = c.touched & 2u;
            pd.hasJump = !cs.hasJumped;
            pd.hasFlip = !cs.hasDoubleJumped && !cs.hasFlipped && cs.airTimeSinceJump < 1.25f;
        }
    }

    uint32_t* recRing = nullptr;   // PINNED host memory, recCap steps of room: the ring's download (45 MB per iteration at BASELINE configs[1]) runs at the link's rate, not at a staged copy's
    std::vector<uint32_t> recResets; std::vector<int32_t> hActsAll, hDoneAll; std::vector<float> hRewAll;
    int recWords = 0, recCap = 0;
    void EnableStepRecords(int tCap) {
        EnvCheck(rlgpu_env_enable_step_records(env, tCap), "enable_step_records");
        recWords = rlgpu_env_step_record_words(env); recCap = tCap;
        if (recRing) { (void)hipHostFree(recRing); recRing = nullptr; }
        if (tCap > 0) HOST_HIP(hipHostMalloc((void**)&recRing, (size_t)tCap * nEnvs * recWords * 4, hipHostMallocDefault));
    }
    // After a fused collection launch of `tUsed` steps at most (env e made steps[e] of them; rows are time-major [t][agent] with `stride` agents per step):
    // every env's steps in order on the worker threads.  With plan.hostReward the rewards come from Match::GetRewards and are written to `rew` (device);
    // otherwise the device's rewards are handed to perEnv.  perEnv(env, StepResult&) as in Step().
    template <class F>
    void ReplayCollected(int tUsed, const int32_t* steps, const int32_t* acts, float* rew, const int32_t* done, F perEnv) {
        const int P = nPlayers; const size_t W = (size_t)recWords, TN = (size_t)tUsed * nAgents;
        if (tUsed <= 0) return;
        if (tUsed > recCap) RG_ERR_CLOSE("deferred host plugins: " << tUsed << " steps collected, the record ring holds " << recCap);
        recResets.resize((size_t)tUsed * nEnvs * (W + 2));
        int nResets = 0;
        EnvCheck(rlgpu_env_download_step_records(env, tUsed, recRing, recResets.data(), tUsed * nEnvs, &nResets), "download_step_records");
        hActsAll.resize(TN); hDoneAll.resize(TN); hRewAll.resize(TN);
        HOST_HIP(hipMemcpy(hActsAll.data(), acts, TN * 4, hipMemcpyDeviceToHost));
        HOST_HIP(hipMemcpy(hDoneAll.data(), done, TN * 4, hipMemcpyDeviceToHost));
        if (!plan.hostReward) HOST_HIP(hipMemcpy(hRewAll.data(), rew, TN * 4, hipMemcpyDeviceToHost));
        // the reset records of every env, by step
        std::vector<std::vector<std::pair<int, const uint32_t*>>> resetsOf((size_t)nEnvs);
        for (int i = 0; i < nResets; i++) {
            const uint32_t* r = recResets.data() + (size_t)i * (W + 2);
            if (r[0] >= (uint32_t)nEnvs || r[1] >= (uint32_t)tUsed) RG_ERR_CLOSE("reset record " << i << " names env " << r[0] << " step " << r[1]);
            resetsOf[r[0]].push_back({(int)r[1], r + 2});
        }
        ForEnvs(nullptr, [&](int e, int, int) {
            RLGSC::Match* M = envMatch[e];
            auto& mine = resetsOf[(size_t)e];
            std::sort(mine.begin(), mine.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            size_t nextReset = 0;
            RLGSC::Gym::StepResult sr;
            for (int t = 0; t < steps[e]; t++) {
                const uint32_t* rec = recRing + ((size_t)t * nEnvs + e) * W;
                FillGameState(sr.state, rec, prevGs[e].lastTickCount, false);
                const int32_t* a = hActsAll.data() + (size_t)t * nAgents + (size_t)e * P;
                if (plan.AnyHost()) M->prevActions = M->ParseActions(RLGSC::IList(a, a + P), prevGs[e]);   // (a callback-only run shares one plugin set: nothing of it is touched)
                sr.done = hDoneAll[(size_t)t * nAgents + (size_t)e * P] != 0;
                if (plan.hostReward) {
                    sr.reward = M->GetRewards(sr.state, sr.done);
                    if ((int)sr.reward.size() != P) RG_ERR_CLOSE("RewardFunction::GetAllRewards returned " << sr.reward.size() << " rewards for " << P << " players");
                    std::copy(sr.reward.begin(), sr.reward.end(), hRewAll.begin() + (size_t)t * nAgents + (size_t)e * P);
                } else sr.reward.assign(hRewAll.begin() + (size_t)t * nAgents + (size_t)e * P, hRewAll.begin() + (size_t)t * nAgents + (size_t)(e + 1) * P);
                prevGs[e] = sr.state;
                perEnv(e, sr);
                if (sr.done) {   // GameInst::Step: gym->Reset() (GameInst.cpp:27-32) -- the kernel did the arena's part inside the step; the plugins' Reset hooks get the new episode's first GameState
                    if (nextReset >= mine.size() || mine[nextReset].first != t) RG_ERR_CLOSE("deferred host plugins: no reset record for env " << e << " step " << t);
                    RLGSC::GameState gs0;
                    FillGameState(gs0, mine[nextReset++].second, 0, true);
                    M->EpisodeReset(gs0);
                    prevGs[e] = gs0;
                }
            }
        });
        if (plan.hostReward) HOST_HIP(hipMemcpy(rew, hRewAll.data(), TN * 4, hipMemcpyHostToDevice));
    }

    // One step of every game with host work in it.  The policy's actions are on the device (`acts`, one per agent); the step's rewards and
    // terminals end up in `rew` / `done`, the observations every player acts on next in `nextObs` (all device rows, one per agent).
    // perEnv(env, StepResult&) runs on the worker threads once the env's StepResult is complete (GameInst's bookkeeping and the step callback).
    template <class F>
    void Step(const int32_t* acts, float* nextObs, float* rew, int32_t* done, F perEnv) {
        const int P = nPlayers;
        const bool plugins = plan.AnyHost();
        if (plugins) HOST_HIP(hipMemcpy(hActs.data(), acts, (size_t)nAgents * 4, hipMemcpyDeviceToHost));
        if (plan.hostParser) {
            ForEnvs(nullptr, [&](int e, int, int) {
                RLGSC::Match* M = envMatch[e];
                M->prevActions = M->ParseActions(RLGSC::IList(hActs.begin() + (size_t)e * P, hActs.begin() + (size_t)(e + 1) * P), prevGs[e]);
                if ((int)M->prevActions.size() != P) RG_ERR_CLOSE("ActionParser::ParseActions returned " << M->prevActions.size() << " actions for " << P << " players");
                for (int k = 0; k < P; k++) for (int j = 0; j < 8; j++) hControls[((size_t)e * P + k) * 8 + j] = M->prevActions[k][j];
            });
            HOST_HIP(hipMemcpy(devControls, hControls.data(), hControls.size() * 4, hipMemcpyHostToDevice));
            EnvCheck(rlgpu_env_step_controls(env, devControls, plan.hostObs ? devObs : nextObs, rew, done), "step_controls");
        } else {
            EnvCheck(rlgpu_env_step(env, acts, plan.hostObs ? devObs : nextObs, rew, done), "step");
        }
        // every env's arena as it stood where Gym::Step builds the GameState (after the first tick and the event tracker)
        EnvCheck(rlgpu_env_download_snapshots(env, snaps.data(), 0, nEnvs), "download_snapshots");
        HOST_HIP(hipMemcpy(hRew.data(), rew, (size_t)nAgents * 4, hipMemcpyDeviceToHost));
        HOST_HIP(hipMemcpy(hDone.data(), done, (size_t)nAgents * 4, hipMemcpyDeviceToHost));
        if (plan.hostObs && hObs.size() < (size_t)nAgents * D) hObs.resize((size_t)nAgents * D);
        ForEnvs(nullptr, [&](int e, int, int) {
            RLGSC::Match* M = envMatch[e];
            RLGSC::Gym::StepResult sr;
            // deltaTickCount: ticks since this env's previous GameState (tickSkip, or 1 on the first step of an episode)
            sr.state = RLGSC::GameState(snaps[e], (int)((uint64_t)snaps[e].tick_count - prevGs[e].lastTickCount));
            if (plugins && !plan.hostParser) M->prevActions = M->ParseActions(RLGSC::IList(hActs.begin() + (size_t)e * P, hActs.begin() + (size_t)(e + 1) * P), prevGs[e]);
            if (plan.hostObs) {
                sr.obs = M->BuildObservations(sr.state);
                for (int k = 0; k < P; k++) {
                    if ((int)sr.obs[k].size() != D) RG_ERR_CLOSE("OBSBuilder::BuildOBS returned " << sr.obs[k].size() << " values, the first observation had " << D);
                    std::copy(sr.obs[k].begin(), sr.obs[k].end(), hObs.begin() + ((size_t)e * P + k) * D);
                }
            }
            sr.done = plan.hostTerminal ? M->IsDone(sr.state) : hDone[(size_t)e * P] != 0;
            if (plan.hostReward) {
                sr.reward = M->GetRewards(sr.state, sr.done);
                if ((int)sr.reward.size() != P) RG_ERR_CLOSE("RewardFunction::GetAllRewards returned " << sr.reward.size() << " rewards for " << P << " players");
                std::copy(sr.reward.begin(), sr.reward.end(), hRew.begin() + (size_t)e * P);
            } else sr.reward.assign(hRew.begin() + (size_t)e * P, hRew.begin() + (size_t)(e + 1) * P);
            for (int k = 0; k < P; k++) hDone[(size_t)e * P + k] = sr.done ? 1 : 0;
            prevGs[e] = sr.state;
            perEnv(e, sr);
        });
        if (plan.hostReward) HOST_HIP(hipMemcpy(rew, hRew.data(), (size_t)nAgents * 4, hipMemcpyHostToDevice));
        if (plan.hostTerminal) HOST_HIP(hipMemcpy(done, hDone.data(), (size_t)nAgents * 4, hipMemcpyHostToDevice));
        if (plan.hostObs) HOST_HIP(hipMemcpy(nextObs, hObs.data(), (size_t)nAgents * D * 4, hipMemcpyHostToDevice));
        if (plugins) {
            std::vector<int32_t> ended;
            for (int e = 0; e < nEnvs; e++) if (hDone[(size_t)e * P]) ended.push_back(e);
            ResetEnvs(ended, nextObs, deviceResets);   // (host_resets: the kernel left the ended envs as they ended -- unless deviceResets)
        } else {
            // a step callback only: the kernel reset the ended envs itself; their next GameState starts a new tick window
            for (int e = 0; e < nEnvs; e++) if (hDone[(size_t)e * P]) prevGs[e].lastTickCount = (uint64_t)snaps[e].tick_count + (uint64_t)(tickSkip - 1);
        }
    }
};

}  // namespace RLGPC
