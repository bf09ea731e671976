// SkillTracker.hip -- RLGPC::SkillTracker (PRIV/Util/SkillTracker.cpp:28-291) on one small device env batch.
// Per step: the current policy and every stored version in use pick deterministic actions for all eval rows (rlgpu_policy_act),
// the host picks per player which policy's action counts (team and teamSwap of its game), the batch steps.  A goal is what the reference
// takes for one (SkillTracker.cpp:129-146): the ball of the step's GameState -- the arena one tick into the step, which the step kernel
// keeps as the env's snapshot (rlgpu_env_enable_snapshots) -- lies behind a goal line, and the policy playing blue scored when its y > 0.
// Like there, a match without a GoalScoreCondition rates every step the ball stays behind the line.  Rewards play no role (the reference
// swaps in a zero reward, :10-16,51).  A user StateSetter without a device form runs on the host for the envs whose episode ended.
#include <hip/hip_runtime.h>

#include <RLGymPPO_CPP/Util/SkillTracker.h>
#include "../../include/rlgpu_state.h"
#include "host_util.h"
#include "HostEnvPath.h"
#include <RLGymSim_CPP/Utils/StateSetters/KickoffState.h>

#include <thread>

namespace RLGPC {

#define SKILL_HIP(call)                                                                                    \
    do {                                                                                                   \
        hipError_t _e = (call);                                                                            \
        if (_e != hipSuccess) RG_ERR_CLOSE("SkillTracker: " #call " failed: " << hipGetErrorString(_e));   \
    } while (0)

struct SkillTracker::Impl {
    rlgpu_learner* cur = nullptr;
    rlgpu_env* env = nullptr;
    std::vector<rlgpu_learner*> old;            // stored versions: inference-only learner objects
    RLGSC::Match* match = nullptr; RLGSC::Gym* gym = nullptr;
    int obsSize = 0, actionAmount = 0, nPlayers = 0, nRows = 0, tickSkip = 8;
    IList layers;
    float *obs = nullptr, *obsNext = nullptr, *rew = nullptr, *logp = nullptr; int32_t *acts = nullptr, *done = nullptr;
    std::mt19937 rng;
    std::vector<GameInst> gameInsts;
    // the eval match uses an obs builder / terminal condition / action parser without a device form: the eval batch steps through the same host
    // path as the training batch does then (HostEnvPath.h; until round 5 the tracker switched itself off instead)
    bool hostPlugins = false; HostEnvPath hp;
    bool hostSetter = false; RLGSC::Arena* scratch = nullptr;   // a user state setter: run on the host facade (Match::ResetState), then uploaded
    std::vector<RlgpuArenaState> snaps, fresh;
    // Gym::Reset of the listed envs with the user's state setter (the kernel has already reset them with the stand-in kickoff setter)
    void HostSetterReset(const std::vector<int32_t>& ids, float* obsRows) {
        if (ids.empty()) return;
        fresh.resize(ids.size());
        EnvCheck(rlgpu_env_download_states(env, fresh.data(), ids.data(), (int)ids.size()), "download_states");
        for (size_t i = 0; i < ids.size(); i++) {
            scratch->_state = fresh[i]; scratch->_SyncFromState();
            (void)match->ResetState(scratch);
            scratch->_SyncToState();
            fresh[i] = scratch->_state;
        }
        EnvCheck(rlgpu_env_upload_states(env, fresh.data(), ids.data(), (int)ids.size()), "upload_states");
        EnvCheck(rlgpu_env_reset_envs(env, ids.data(), (int)ids.size(), 0, obsRows), "reset_envs");
    }
    void ResetAll(float* obsRows) {
        if (hostPlugins) {
            EnvCheck(rlgpu_env_reset(env, 1, hp.plan.hostObs ? hp.devObs : obsRows), "reset");
            std::vector<int32_t> all(gameInsts.size()); for (size_t e = 0; e < all.size(); e++) all[e] = (int32_t)e;
            hp.ResetEnvs(all, obsRows, true);
            return;
        }
        EnvCheck(rlgpu_env_reset(env, 1, obsRows), "reset");
        if (hostSetter) { std::vector<int32_t> all(gameInsts.size()); for (size_t e = 0; e < all.size(); e++) all[e] = (int32_t)e; HostSetterReset(all, obsRows); }
    }
    void EnvCheck(int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("SkillTracker: rlgpu_env_" << what << " failed (" << rc << "): " << rlgpu_env_last_error(env)); }
    void LrnCheck(rlgpu_learner* l, int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("SkillTracker: rlgpu_" << what << " failed (" << rc << "): " << rlgpu_learner_last_error(l)); }
    rlgpu_learner* MakeVersion(const float* params) {
        RlgpuLearnerConfig lc{};
        lc.obs_size = obsSize; lc.n_actions = actionAmount; lc.n_policy_layers = (int)layers.size(); lc.n_critic_layers = 1;
        for (int i = 0; i < 8; i++) { lc.policy_layers[i] = 16; lc.critic_layers[i] = 16; }
        for (size_t i = 0; i < layers.size(); i++) lc.policy_layers[i] = layers[i];
        lc.clip_range = 0.2f; lc.temperature = 1; lc.max_rows = std::max(nRows, 64);
        rlgpu_learner* l = nullptr;
        int rc = rlgpu_learner_create(&l, 0, &lc);
        LrnCheck(l, rc, "learner_create");
        LrnCheck(l, rlgpu_learner_set_params(l, 0, params), "learner_set_params");
        return l;
    }
    void ResetGame(Game& g, int numPolicies) {   // SkillTracker.h:25-28
        g.teamSwap = std::uniform_real_distribution<float>(0, 1)(rng) > 0.5f;
        g.oldPolicyIndex = numPolicies > 0 ? (int)(rng() % (unsigned)numPolicies) : 0;
    }
};

SkillTracker::SkillTracker(const SkillTrackerConfig& config_, rlgpu_learner* learner, int obsSize, int actionAmount, const IList& policyLayerSizes, int randomSeed,
                           RenderSender* renderSender_)
    : renderSender(renderSender_), config(config_), impl(new Impl()) {
    Impl& m = *impl;
    if (config.numEnvs <= 0 || config.timestepsPerVersion < 0 || config.maxVersions <= 0 || config.simTime <= 0 || config.updateInterval <= 0)
        RG_ERR_CLOSE("SkillTracker: numEnvs, maxVersions, simTime and updateInterval must be positive");
    if (!config.envCreateFunc) RG_ERR_CLOSE("SkillTracker: envCreateFunc is NULL");
    m.cur = learner; m.obsSize = obsSize; m.actionAmount = actionAmount; m.layers = policyLayerSizes;
    m.rng.seed((unsigned)randomSeed * 2654435761u + 99u);
    EnvCreateResult ecr = config.envCreateFunc();
    if (!ecr.match || !ecr.gym) RG_ERR_CLOSE("SkillTracker: envCreateFunc returned a null match or gym");
    m.match = ecr.match; m.gym = ecr.gym; m.tickSkip = ecr.gym->tickSkip;
    if (config.kickoffStatesOnly) m.match->stateSetter = new RLGSC::KickoffState();   // SkillTracker.cpp:48-49 (the env's own setter is discarded, not freed)
    RLGSC::Match::DevicePlan plan = m.match->PlanDevice(m.tickSkip);
    // the eval games' reward is the zero reward of SkillTracker.cpp:10-16,51, whatever the match's own function is: nothing to run for it, on either side
    plan.hostReward = false;
    m.hostPlugins = plan.hostTerminal || plan.hostObs || plan.hostParser;
    if (m.hostPlugins)
        RG_LOG("SkillTracker: the eval match uses " << (plan.hostTerminal ? "a terminal condition " : "") << (plan.hostObs ? "an obs builder " : "") << (plan.hostParser ? "an action parser " : "")
               << "without a device form: its games step through the host path (the plugins run on the host every step, the arenas stay on the device)");
    m.hostSetter = plan.hostSetter && !m.hostPlugins;   // (with host plugins the host path runs the user's setter too)
    plan.cfg.host_resets = (m.hostPlugins || plan.hostSetter) ? 1 : 0;   // (a host REWARD alone asks for nothing here)
    RlgpuGymConfig g = plan.cfg;
    g.n_terms = 0; g.zero_sum = 0;                                     // the zero reward of SkillTracker.cpp:10-16,51
    for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) g.event_weights[i] = 0.f;
    g.seed_lo = (uint32_t)randomSeed + 7777u; g.seed_hi = 1;
    int rc = rlgpu_env_create(&m.env, 0, config.numEnvs, m.match->teamSize, &g);
    m.EnvCheck(rc, "create");
    if (!m.gym->arena->GetMutatorConfig().IsDefault()) { const RlgpuMutators mut = m.gym->arena->GetMutatorConfig().ToDevice(); m.EnvCheck(rlgpu_env_set_mutators(m.env, &mut), "set_mutators"); }
    std::filesystem::path soccar = RocketSim::GetCollisionMeshFolder() / "soccar";
    if (!RocketSim::GetCollisionMeshFolder().empty() && std::filesystem::is_directory(soccar)) m.EnvCheck(rlgpu_env_load_cmf_dir(m.env, soccar.string().c_str()), "load_cmf_dir");
    else m.EnvCheck(rlgpu_env_set_procedural_mesh(m.env), "set_procedural_mesh");
    if (!plan.hostObs && rlgpu_env_obs_size(m.env) != obsSize) RG_ERR_CLOSE("SkillTracker: the eval env's observations have " << rlgpu_env_obs_size(m.env) << " values, the policy takes " << obsSize);
    m.nPlayers = m.match->playerAmount; m.nRows = rlgpu_env_num_agents(m.env);
    SKILL_HIP(hipMalloc(&m.obs, (size_t)m.nRows * obsSize * 4)); SKILL_HIP(hipMalloc(&m.obsNext, (size_t)m.nRows * obsSize * 4));
    SKILL_HIP(hipMalloc(&m.rew, m.nRows * 4)); SKILL_HIP(hipMalloc(&m.logp, m.nRows * 4));
    SKILL_HIP(hipMalloc(&m.acts, m.nRows * 4)); SKILL_HIP(hipMalloc(&m.done, m.nRows * 4));
    m.EnvCheck(rlgpu_env_enable_snapshots(m.env, 1), "enable_snapshots");
    if (m.hostSetter) m.scratch = RLGSC::MakeScratchArena(m.match->teamSize, m.match->spawnOpponents);
    m.gameInsts.resize(config.numEnvs);
    if (m.hostPlugins) {
        if (config.kickoffStatesOnly) {   // every env's own plugin set plays kickoffs too (SkillTracker.cpp:48-49)
            const EnvCreateFn inner = config.envCreateFunc;
            config.envCreateFunc = [inner]() { EnvCreateResult r = inner(); if (r.match) r.match->stateSetter = new RLGSC::KickoffState(); return r; };
        }
        m.hp.Setup(m.env, plan, m.match, m.gym, config.envCreateFunc, (int)std::thread::hardware_concurrency(), config.numEnvs, m.nPlayers, m.nRows, obsSize, rlgpu_env_obs_size(m.env), m.tickSkip);
    }
    m.ResetAll(m.obs);

    modeName = std::to_string(m.match->teamSize) + "v" + std::to_string(m.match->teamSize);   // ModeNameFromGameInst, SkillTracker.cpp:20-26
    if (config.perModeRatings) { modeNames.insert(modeName); curRating.data[modeName] = config.initialRating; }
    else { modeName = ""; curRating.data[""] = config.initialRating; }
    games.resize(config.numEnvs);
    for (Game& game : games) m.ResetGame(game, 1);
    for (int e = 0; e < config.numEnvs; e++) { m.gameInsts[e].gym = m.gym; m.gameInsts[e].match = m.match; m.gameInsts[e].index = e; m.gameInsts[e].isEval = true; }
}

SkillTracker::~SkillTracker() {
    Impl& m = *impl;
    for (void* p : {(void*)m.obs, (void*)m.obsNext, (void*)m.rew, (void*)m.logp, (void*)m.acts, (void*)m.done}) if (p) (void)hipFree(p);
    for (rlgpu_learner* l : m.old) rlgpu_learner_destroy(l);
    if (m.env) rlgpu_env_destroy(m.env);
    delete m.scratch;
    delete m.gym; delete m.match;
    delete impl;
}

int SkillTracker::NumOldPolicies() const { return (int)impl->old.size(); }

void SkillTracker::AppendOldPolicy(const std::vector<float>& policyParams, RatingSet rating) {
    impl->old.push_back(impl->MakeVersion(policyParams.data()));
    oldRatings.push_back(rating);
}

void SkillTracker::UpdateRatings(RatingSet& winner, RatingSet& loser, bool updateWinner, bool updateLoser, std::string mode) {
    // simple elo step per goal (SkillTracker.cpp:72-86)
    if (!winner.data.count(mode) || !loser.data.count(mode)) RG_ERR_CLOSE("SkillTracker::UpdateRatings(): no rating for mode \"" << mode << "\"");
    const float expDelta = (loser.data[mode] - winner.data[mode]) / 400;
    const float expected = 1 / (powf(10, expDelta) + 1);
    if (updateWinner) winner.data[mode] += config.ratingInc * (1 - expected);
    if (updateLoser) loser.data[mode] += config.ratingInc * (expected - 1);
}

void SkillTracker::RunGames(int64_t timestepsDelta) {
    Impl& m = *impl;
    if (runCounter++ % (uint64_t)config.updateInterval != 0) return;

    auto snapshot = [&]() {   // the current policy's parameters as a new stored version
        std::vector<float> params((size_t)rlgpu_learner_num_params(m.cur, 0));
        m.LrnCheck(m.cur, rlgpu_learner_get_params(m.cur, 0, params.data()), "learner_get_params");
        AppendOldPolicy(params, curRating);
    };
    if (m.old.empty() && config.startWithVersion) snapshot();

    if (!m.old.empty()) {
        const RatingSet prevRating = curRating;
        const float timePerGame = config.simTime / (float)games.size();
        const int numSteps = (int)(timePerGame * 120 / m.tickSkip);
        if (numSteps <= 0) RG_ERR_CLOSE("RLGPC::SkillTracker RunGames(): simTime is too low for the number of games, there is not enough time per game to step");
        const int nOld = (int)m.old.size();
        for (Game& g : games) if (g.oldPolicyIndex >= nOld) g.oldPolicyIndex = nOld - 1;
        std::vector<int32_t> picksCur(m.nRows), picks(m.nRows), dones(m.nRows);
        std::vector<std::vector<int32_t>> picksOld(nOld, std::vector<int32_t>(m.nRows));
        std::vector<float> rews(m.nRows);
        const StepCallback& cb = config.stepCallback;
        std::vector<RlgpuArenaState>& states = m.snaps;   // every env's arena where this step's GameState was taken
        states.resize(games.size());
        std::vector<int32_t> ended;
        for (int s = 0; s < numSteps; s++) {
            m.LrnCheck(m.cur, rlgpu_policy_act(m.cur, m.obs, m.nRows, 1, nullptr, m.acts, m.logp), "policy_act");
            m.LrnCheck(m.cur, rlgpu_learner_sync(m.cur), "learner_sync");
            SKILL_HIP(hipMemcpy(picksCur.data(), m.acts, m.nRows * 4, hipMemcpyDeviceToHost));
            std::vector<char> used(nOld, 0);
            for (const Game& g : games) used[g.oldPolicyIndex] = 1;
            for (int k = 0; k < nOld; k++) {
                if (!used[k]) continue;
                m.LrnCheck(m.old[k], rlgpu_policy_act(m.old[k], m.obs, m.nRows, 1, nullptr, m.acts, m.logp), "policy_act");
                m.LrnCheck(m.old[k], rlgpu_learner_sync(m.old[k]), "learner_sync");
                SKILL_HIP(hipMemcpy(picksOld[k].data(), m.acts, m.nRows * 4, hipMemcpyDeviceToHost));
            }
            // blue plays the current policy unless the game is swapped (SkillTracker.cpp:104-127); cars alternate blue, orange, blue, ...
            for (size_t e = 0; e < games.size(); e++)
                for (int j = 0; j < m.nPlayers; j++) {
                    const bool blue = (j % 2) == 0, curPlays = blue != games[e].teamSwap;
                    const size_t row = e * m.nPlayers + j;
                    picks[row] = curPlays ? picksCur[row] : picksOld[games[e].oldPolicyIndex][row];
                }
            SKILL_HIP(hipMemcpy(m.acts, picks.data(), m.nRows * 4, hipMemcpyHostToDevice));
            if (m.hostPlugins) {
                // Gym::Step with the user's plugins on the host (HostEnvPath::Step: parser, obs builder, terminal conditions; the ended games are reset
                // there, by the user's state setter or the device's); what this loop needs comes back in hp.snaps / hp.hRew / hp.hDone
                m.hp.Step(m.acts, m.obsNext, m.rew, m.done, [](int, RLGSC::Gym::StepResult&) {});
                std::swap(m.obs, m.obsNext);
                rews = m.hp.hRew; dones = m.hp.hDone; states = m.hp.snaps;
            } else {
            m.EnvCheck(rlgpu_env_step(m.env, m.acts, m.obsNext, m.rew, m.done), "step");
            m.EnvCheck(rlgpu_env_sync(m.env), "sync");
            std::swap(m.obs, m.obsNext);
            SKILL_HIP(hipMemcpy(rews.data(), m.rew, m.nRows * 4, hipMemcpyDeviceToHost));
            SKILL_HIP(hipMemcpy(dones.data(), m.done, m.nRows * 4, hipMemcpyDeviceToHost));
            m.EnvCheck(rlgpu_env_download_snapshots(m.env, states.data(), 0, (int)games.size()), "download_snapshots");
            }
            ended.clear();
            for (size_t e = 0; e < games.size(); e++) {
                Game& g = games[e];
                const float ballY = states[e].ball.pos[1];
                if (RLGSC::Math::IsBallScored(Vec(states[e].ball.pos[0], ballY, states[e].ball.pos[2]))) {   // SkillTracker.cpp:129-146
                    const bool blueScored = ballY > 0;
                    const bool curScored = blueScored != g.teamSwap;
                    if (curScored) UpdateRatings(curRating, oldRatings[g.oldPolicyIndex], true, true, modeName);
                    else UpdateRatings(oldRatings[g.oldPolicyIndex], curRating, true, true, modeName);
                }
                if (cb) {
                    RLGSC::Gym::StepResult sr;
                    sr.state = RLGSC::GameState(states[e], m.tickSkip);
                    sr.reward.assign(rews.begin() + e * m.nPlayers, rews.begin() + (e + 1) * m.nPlayers);
                    sr.done = dones[e * m.nPlayers] != 0;
                    m.gameInsts[e].totalSteps++;
                    cb(&m.gameInsts[e], sr, m.gameInsts[e]._metrics);
                }
                if (dones[e * m.nPlayers]) { m.ResetGame(g, nOld); ended.push_back((int32_t)e); }
            }
            if (m.hostSetter) m.HostSetterReset(ended, m.obs);
            if (renderSender) {   // SkillTracker.cpp:147-151
                RLGSC::GameState gs(states[0], m.tickSkip);
                renderSender->Send(gs, m.match->actionParser->ParseActions(RLGSC::IList(picks.begin(), picks.begin() + m.nPlayers), gs));
                if (!std::getenv("RLGPU_RENDER_NO_SLEEP")) std::this_thread::sleep_for(std::chrono::microseconds((int64_t)(m.tickSkip / 120.f * 1e6f)));
            }
        }
        RG_LOG("New ratings:");
        for (auto& pair : curRating.data) {
            auto it = prevRating.data.find(pair.first);
            if (it == prevRating.data.end()) continue;
            const float delta = pair.second - it->second;
            RG_LOG(" > " << pair.first << (pair.first.empty() ? "" : " ") << std::setprecision(6) << pair.second << " (" << (delta >= 0 ? "+" : "") << std::setprecision(4) << delta << ")");
        }
    } else {
        RG_LOG(" > No old policies yet, skipping");
    }

    timestepsSinceVersionMade += timestepsDelta;
    if (timestepsSinceVersionMade >= config.timestepsPerVersion) {   // SkillTracker.cpp:237-256
        m.ResetAll(m.obs);
        timestepsSinceVersionMade = 0;
        snapshot();
        if ((int)m.old.size() > config.maxVersions) {
            rlgpu_learner_destroy(m.old[0]);
            m.old.erase(m.old.begin());
            oldRatings.erase(oldRatings.begin());
        }
    }
}

// ---- "skill_rating" in RUNNING_STATS.json (Learner.cpp:185-194; SkillTracker.cpp:259-291) ----
std::string SkillTracker::RatingsToJSON() const {
    std::ostringstream o;
    o << std::setprecision(9);
    if (!config.perModeRatings) { auto it = curRating.data.find(""); o << (it == curRating.data.end() ? config.initialRating : it->second); return o.str(); }
    o << "{";
    bool first = true;
    for (auto& pair : curRating.data) { o << (first ? "" : ", ") << "\"" << pair.first << "\": " << pair.second; first = false; }
    o << "}";
    return o.str();
}

SkillTracker::RatingSet SkillTracker::LoadRatingSet(const std::string& jsonValue, bool warn) {
    constexpr const char* ERR_PREFIX = "RLGPC::SkillTracker::LoadRatingSet(): ";
    RatingSet result;
    size_t i = 0;
    while (i < jsonValue.size() && isspace((unsigned char)jsonValue[i])) i++;
    if (i < jsonValue.size() && jsonValue[i] == '{') {
        if (config.perModeRatings) {
            // "mode": number pairs
            while ((i = jsonValue.find('"', i)) != std::string::npos) {
                const size_t e = jsonValue.find('"', i + 1);
                if (e == std::string::npos) break;
                const std::string key = jsonValue.substr(i + 1, e - i - 1);
                const size_t c = jsonValue.find(':', e);
                if (c == std::string::npos) break;
                result.data[key] = std::stof(jsonValue.substr(c + 1));
                i = c + 1;
            }
            for (auto& mode : modeNames)
                if (!result.data.count(mode)) {
                    if (warn) RG_LOG(ERR_PREFIX << "Loaded ratings are missing mode \"" << mode << "\", rating will be set to initial rating.");
                    result.data[mode] = config.initialRating;
                }
        } else {
            if (warn) RG_LOG(ERR_PREFIX << "Loaded ratings are per-mode, but per-mode ratings are disabled. Rating will be set to initial rating.");
            result.data[""] = config.initialRating;
        }
    } else {
        const float v = std::stof(jsonValue.substr(i));
        if (config.perModeRatings) {
            if (warn) RG_LOG(ERR_PREFIX << "Loaded ratings are not per-mode, all mode ratings will be set to the loaded rating: " << v);
            for (auto& mode : modeNames) result.data[mode] = v;
        } else {
            result.data[""] = v;
        }
    }
    return result;
}

}  // namespace RLGPC
