// GPU check of the open plugin boundary (tests/test_host_cpp.py runs it): user subclasses of RewardFunction, OBSBuilder,
// TerminalCondition, StateSetter and ActionParser -- written against include/ exactly as against the reference's headers -- run on the
// host every step while the arenas stay on the device; the standalone Gym steps one arena through the same kernel.
//   part 1  host forms == device forms: the built-in stack forced onto the host gives the device path's experience
//   part 2  user plugins: rewards / dones / observations / resets are what the plugins compute, step callbacks see them, training runs
//   part 3  Gym::Reset / Gym::Step / Arena facade
#include <RLGymPPO_CPP/Learner.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CombinedReward.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/ZeroSumReward.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/StateSetters/RandomState.h>
#include <RLGymSim_CPP/Utils/StateSetters/KickoffState.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cmath>

using namespace RLGPC;
using namespace RLGSC;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

// ---- part 1: the built-ins with their device form switched off ---------------------------------------------------------------------
struct HostOnlyReward : CombinedReward {
    using CombinedReward::CombinedReward;
    bool AddDeviceTerms(RlgpuGymConfig&, float) const override { return false; }
};
struct HostOnlyZeroSum : ZeroSumReward {
    using ZeroSumReward::ZeroSumReward;
    bool AddDeviceTerms(RlgpuGymConfig&, float) const override { return false; }
};
struct HostOnlyNoTouch : NoTouchCondition {
    using NoTouchCondition::NoTouchCondition;
    bool AddDeviceCondition(RlgpuGymConfig&) const override { return false; }
};
struct HostOnlyOBS : DefaultOBS {
    bool ApplyToDevice(RlgpuGymConfig&) const override { return false; }
};
struct HostOnlyParser : DiscreteAction {
    bool ApplyToDevice(RlgpuGymConfig&) const override { return false; }
};

static int g_teamSize = 1;
static bool g_host = false, g_hostParser = false, g_spawnOpponents = true;
static EnvCreateResult MakeBuiltinEnv() {
    auto terms = std::vector<std::pair<RewardFunction*, float>>{
        {new FaceBallReward(), 0.1f}, {new VelocityPlayerToBallReward(), 0.5f}, {new VelocityBallToGoalReward(), 1.0f}, {new TouchBallReward(0.5f), 2.f},
        {new SaveBoostReward(), 0.3f}, {new VelocityReward(), 0.2f}, {new EventReward({.teamGoal = 1.f, .concede = -1.f, .touch = 0.5f, .boostPickup = 0.25f}), 10.f}};
    RewardFunction* reward;
    if (g_teamSize > 1) reward = g_host ? (RewardFunction*)new HostOnlyZeroSum(new CombinedReward(terms), 0.4f, 0.8f) : new ZeroSumReward(new CombinedReward(terms), 0.4f, 0.8f);
    else reward = g_host ? (RewardFunction*)new HostOnlyReward(terms) : new CombinedReward(terms);
    std::vector<TerminalCondition*> terminal = {g_host ? (TerminalCondition*)new HostOnlyNoTouch(12) : new NoTouchCondition(12), new GoalScoreCondition()};
    OBSBuilder* obs = g_host ? (OBSBuilder*)new HostOnlyOBS() : new DefaultOBS();
    ActionParser* parser = g_hostParser ? (ActionParser*)new HostOnlyParser() : new DiscreteAction();
    Match* match = new Match(reward, terminal, obs, parser, new RandomState(true, true, false), g_teamSize, g_spawnOpponents);
    return {match, new Gym(match, 8)};
}

static LearnerConfig SmallConfig(int envs, int steps, int players) {
    LearnerConfig cfg = {};
    cfg.numThreads = 4; cfg.numGamesPerThread = envs / 4;
    cfg.timestepsPerIteration = (int64_t)envs * players * steps;
    cfg.ppo.batchSize = cfg.timestepsPerIteration; cfg.ppo.miniBatchSize = cfg.timestepsPerIteration; cfg.expBufferSize = cfg.timestepsPerIteration;
    cfg.ppo.epochs = 1; cfg.ppo.policyLayerSizes = {64, 64}; cfg.ppo.criticLayerSizes = {64, 64};
    cfg.ppo.policyLR = cfg.ppo.criticLR = 3e-4f;
    cfg.randomSeed = 77;
    cfg.sendMetrics = false; cfg.checkpointSaveFolder = ""; cfg.checkpointLoadFolder = "";
    cfg.lockstepCollection = true;   // these checks compare paths row by row: every game makes the same number of steps
    return cfg;
}

struct Collected { std::vector<float> obs, rew; std::vector<int32_t> acts, done; int D = 0, T = 0, agents = 0; };
static Collected CollectOnce(bool host, bool hostParser, int teamSize, int envs, int steps, int iterations = 1, int players = 0) {
    g_host = host; g_hostParser = hostParser; g_teamSize = teamSize;
    Learner learner(MakeBuiltinEnv, SmallConfig(envs, steps, players ? players : 2 * teamSize));
    Collected c;
    for (int i = 0; i < iterations; i++) learner.CollectTimesteps();
    learner.CopyCollected(&c.obs, &c.acts, &c.rew, &c.done);
    c.D = learner.obsSize; c.T = learner.StepsPerIteration(); c.agents = learner.NumAgents();
    return c;
}

static int ComparePaths(int teamSize, bool hostParser, bool spawnOpponents = true) {
    const int envs = 32, steps = 48;
    g_spawnOpponents = spawnOpponents;
    const int players = spawnOpponents ? 2 * teamSize : teamSize;
    Collected dev = CollectOnce(false, false, teamSize, envs, steps, 2, players), host = CollectOnce(true, hostParser, teamSize, envs, steps, 2, players);
    g_spawnOpponents = true;
    CHECK(dev.agents == envs * players && dev.D == 51 + 19 * players);
    CHECK(dev.D == host.D && dev.T == host.T && dev.agents == host.agents && dev.T == steps);
    CHECK(dev.acts == host.acts);                       // same observations -> same sampled actions, step after step
    CHECK(dev.done == host.done);
    int dones = 0; for (int32_t d : dev.done) dones += d;
    CHECK(dones > 0);                                   // episodes ended (and restarted through the host path) inside the window
    double worstObs = 0, worstRew = 0, sumAbsRew = 0;
    for (size_t i = 0; i < dev.obs.size(); i++) worstObs = std::max(worstObs, (double)std::fabs(dev.obs[i] - host.obs[i]));
    for (size_t i = 0; i < dev.rew.size(); i++) { worstRew = std::max(worstRew, (double)std::fabs(dev.rew[i] - host.rew[i])); sumAbsRew += std::fabs(dev.rew[i]); }
    std::printf("team size %d%s%s: %d dones, max |obs diff| %.3g, max |reward diff| %.3g (mean |reward| %.3g)\n", teamSize, hostParser ? " + host parser" : "", spawnOpponents ? "" : ", no opponents", dones,
                worstObs, worstRew, sumAbsRew / dev.rew.size());
    CHECK(worstObs <= 1e-6);                            // same arithmetic on both sides: one multiply per value
    CHECK(worstRew <= 2e-5);                            // sums of products: fp32 rounding order only
    return 0;
}

// ---- part 2: plugins only a user would write ----------------------------------------------------------------------------------------
struct BallHeightReward : RewardFunction {                  // stateless custom reward
    float GetReward(const PlayerData& player, const GameState& state, const Action& prev) override {
        return state.ball.pos.z / CommonValues::CEILING_Z + 0.01f * prev.throttle - (player.carState.isOnGround ? 0.f : 0.05f);
    }
};
struct StepLimitCondition : TerminalCondition {             // per-episode state: needs its Reset hook and one instance per env
    int steps = 0, limit;
    StepLimitCondition(int limit) : limit(limit) {}
    void Reset(const GameState&) override { steps = 0; }
    bool IsTerminal(const GameState&) override { return ++steps >= limit; }
};
struct RangeOBS : DefaultOBS {                              // DefaultOBS plus three values of its own
    FList BuildOBS(const PlayerData& player, const GameState& state, const Action& prevAction) override {
        FList obs = DefaultOBS::BuildOBS(player, state, prevAction);
        const Vec d = state.ball.pos - player.phys.pos;
        obs += {d.Length() / 6000.f, (float)state.lastTickCount, (float)player.carId};
        return obs;
    }
};
struct BallOverCarSetter : StateSetter {                    // kickoff, then the ball 300 uu above blue's first car, falling
    GameState ResetState(Arena* arena) override {
        arena->ResetToRandomKickoff();
        Car* first = arena->_cars[0];
        CarState cs = first->GetState();
        cs.boost = 77.f;
        first->SetState(cs);
        BallState bs;
        bs.pos = cs.pos + Vec(0, 0, 300); bs.vel = Vec(0, 0, -100);
        arena->ball->SetState(bs);
        return GameState(arena);
    }
};
struct EightWayParser : ActionParser {                      // 8 actions: throttle x steer x boost
    int GetActionAmount() override { return 8; }
    ActionSet ParseActions(const Input& in, const GameState&) override {
        ActionSet out(in.size());
        for (size_t i = 0; i < in.size(); i++) { out[i].throttle = (in[i] & 1) ? 1.f : -1.f; out[i].steer = (in[i] & 2) ? 1.f : -1.f; out[i].boost = (in[i] & 4) ? 1.f : 0.f; }
        return out;
    }
};

static std::atomic<int> g_envsMade{0};
static EnvCreateResult MakeUserEnv() {
    g_envsMade++;
    Match* match = new Match(new BallHeightReward(), {new StepLimitCondition(5)}, new RangeOBS(), new EightWayParser(), new BallOverCarSetter(), 1, true);
    return {match, new Gym(match, 8)};
}

static std::atomic<int> g_callbacks{0}, g_callbackErrors{0};
static void CheckingCallback(GameInst* game, const Gym::StepResult& r, Report& metrics) {
    g_callbacks++;
    BallHeightReward again;
    for (size_t i = 0; i < r.state.players.size(); i++) {
        const float want = again.GetReward(r.state.players[i], r.state, game->match->prevActions[i]);
        if (std::fabs(want - r.reward[i]) > 1e-6f) g_callbackErrors++;
    }
    if (r.obs.size() != 2 || r.obs[0].size() != 89 + 3) g_callbackErrors++;
    metrics.AccumAvg("ball_height", r.state.ball.pos.z);
}

static int UserPlugins() {
    const int envs = 16, steps = 12;
    LearnerConfig cfg = SmallConfig(envs, steps, 2);
    Learner learner(MakeUserEnv, cfg);
    CHECK(g_envsMade == envs);                          // one plugin set per game, like the reference's GameInsts
    CHECK(learner.obsSize == 89 + 3 && learner.actionAmount == 8);
    learner.stepCallback = CheckingCallback;
    learner.CollectTimesteps();
    Collected c;
    learner.CopyCollected(&c.obs, &c.acts, &c.rew, &c.done);
    const int D = learner.obsSize, N = learner.NumAgents(), T = learner.StepsPerIteration();
    CHECK(T == steps && g_callbacks == envs * steps && g_callbackErrors == 0);
    for (int t = 0; t < T; t++)
        for (int a = 0; a < N; a++) {
            CHECK(c.done[(size_t)t * N + a] == ((t + 1) % 5 == 0 ? 1 : 0));        // StepLimitCondition(5), counted per env on the host
            CHECK(c.acts[(size_t)t * N + a] >= 0 && c.acts[(size_t)t * N + a] < 8);
            const float* row = &c.obs[((size_t)(t + 1) * N + a) * D];
            CHECK(row[D - 1] == (float)(a % 2 + 1));                               // RangeOBS's own values arrive in the policy's input
            const float* prevAct = row + 9;                                        // DefaultOBS: ball (9) | previous action (8) | ...
            if ((t + 1) % 5 == 0) {
                // first observation of the next episode, from BallOverCarSetter: ball 300 uu above blue's first car, no previous action
                const float* blueCar = (a % 2 == 0) ? row + 51 : row + 51 + 19;    // own block for blue, the opponent block for orange
                const float sx = (a % 2 == 0) ? 1.f : -1.f;                        // orange sees the mirrored field
                CHECK(std::fabs(row[0] - blueCar[0]) < 1e-6f && std::fabs(row[1] - blueCar[1]) < 1e-6f);
                CHECK(std::fabs(row[2] * CommonValues::CEILING_Z - (17.f + 300.f)) < 1e-2f);
                CHECK(std::fabs(row[5] * CommonValues::CAR_MAX_SPEED - (-100.f)) < 1e-3f);
                CHECK(std::fabs(blueCar[15] - 0.77f) < 1e-6f);                     // boost fraction set through Car::SetState
                (void)sx;
                for (int j = 0; j < 8; j++) CHECK(prevAct[j] == 0.f);
            } else {
                const int32_t act = c.acts[(size_t)t * N + a];                     // EightWayParser's controls, as the obs builder saw them
                CHECK(prevAct[0] == ((act & 1) ? 1.f : -1.f) && prevAct[1] == ((act & 2) ? 1.f : -1.f) && prevAct[6] == ((act & 4) ? 1.f : 0.f));
            }
        }
    // the rewards in the experience buffer are BallHeightReward's
    double sum = 0; for (float r : c.rew) sum += r;
    CHECK(sum > 0 && std::isfinite(sum));
    std::vector<Report> perGame = learner.GetAllGameMetrics();
    CHECK((int)perGame.size() == envs && perGame[3].Has("ball_height_avg_count"));
    // and it trains: two full iterations with finite statistics
    int iterations = 0; bool finite = true;
    learner.stepCallback = nullptr;
    learner.iterationCallback = [&](Learner* l, Report& report) {
        iterations++;
        finite = finite && std::isfinite(report["Policy Entropy"]) && std::isfinite(report["Value Function Loss"]) && report["Policy Entropy"] > 0;
        if (iterations == 2) l->config.timestepLimit = 1;
    };
    setenv("RLGPU_QUIET", "1", 1);
    const auto t0 = std::chrono::steady_clock::now();
    learner.Learn();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    CHECK(iterations == 2 && finite);
    std::printf("user plugins: %d envs x %d steps per iteration, 2 iterations in %.2f s\n", envs, steps, secs);
    return 0;
}

// ---- part 2c: a user RewardFunction as the only host plugin stays in the fused launch (LearnerConfig::deferHostRewards; VERDICT r05 "next" 3) ----------
// In the reference a user's reward runs inside the agent threads at full speed (ThreadAgent.cpp:100-160, Match.cpp:25-30).  Nothing of it feeds the next
// action, so the Learner keeps the fused collection launch, has the kernel store every step's GameState source (rlgpu_env_enable_step_records) and replays
// each env's steps in order on the host afterwards: Match::GetRewards, the plugins' Reset at episode starts, GameInst's bookkeeping, the step callback.
// The reward below carries per-episode state and reads the previous action and the match counters, so a wrong order, a missing Reset or a wrong initial
// state shows.  Against the per-step host path (deferHostRewards = false): same actions, observations and terminals; rewards to 1e-6.
static std::atomic<int> g_rewardResets{0}, g_rewardPreSteps{0};
struct EpisodeProgressReward : RewardFunction {
    int stepsInEpisode = 0; float startBallZ = 0, lastBallSpeed = 0;
    void Reset(const GameState& initialState) override { g_rewardResets++; stepsInEpisode = 0; startBallZ = initialState.ball.pos.z; lastBallSpeed = initialState.ball.vel.Length(); }
    void PreStep(const GameState& state) override { g_rewardPreSteps++; stepsInEpisode++; }
    float GetReward(const PlayerData& player, const GameState& state, const Action& prev) override {
        const float speed = state.ball.vel.Length();
        const float r = 0.001f * (state.ball.pos.z - startBallZ) + 0.01f * stepsInEpisode + 0.1f * prev.throttle + 0.05f * prev.boost + (player.ballTouchedStep ? 1.f : 0.f)
                        + 0.2f * player.boostFraction + 0.5f * (float)player.matchGoals + 1e-4f * (speed - lastBallSpeed) + (player.carState.isOnGround ? 0.f : -0.02f)
                        + 1e-3f * (float)state.deltaTickCount + (state.boostPads[3] ? 0.003f : 0.f);
        return r;
    }
    float GetFinalReward(const PlayerData& player, const GameState& state, const Action& prev) override { return GetReward(player, state, prev) + 5.f; }
    std::vector<float> GetAllRewards(const GameState& state, const ActionSet& prevActions, bool final) override {
        std::vector<float> out = RewardFunction::GetAllRewards(state, prevActions, final);
        lastBallSpeed = state.ball.vel.Length();
        return out;
    }
};
static int g_deferTeam = 1;
static EnvCreateResult MakeUserRewardEnv() {
    Match* match = new Match(new EpisodeProgressReward(), {new NoTouchCondition(10), new GoalScoreCondition()}, new DefaultOBS(), new DiscreteAction(), new RandomState(true, true, false), g_deferTeam, true);
    return {match, new Gym(match, 8)};
}
static std::atomic<int> g_deferCallbacks{0};
static std::atomic<long long> g_deferCallbackSum{0};
static void DeferCallback(GameInst* game, const Gym::StepResult& r, Report& metrics) {
    g_deferCallbacks++;
    long long v = (long long)r.state.lastTickCount + (r.done ? 1000 : 0);
    for (float x : r.reward) v += (long long)std::llround(x * 1000.0);
    g_deferCallbackSum += v;
    metrics.AccumAvg("ball_height", r.state.ball.pos.z);
}
static int DeferredUserReward(int teamSize) {
    const int envs = 32, steps = 48, players = 2 * teamSize;
    g_deferTeam = teamSize;
    Collected got[2]; int resets[2], presteps[2], callbacks[2]; long long sums[2]; bool fused[2];
    for (int pass = 0; pass < 2; pass++) {
        g_rewardResets = 0; g_rewardPreSteps = 0; g_deferCallbacks = 0; g_deferCallbackSum = 0;
        LearnerConfig cfg = SmallConfig(envs, steps, players);
        cfg.deferHostRewards = pass == 0;
        Learner learner(MakeUserRewardEnv, cfg);
        learner.stepCallback = DeferCallback;
        for (int i = 0; i < 2; i++) learner.CollectTimesteps();
        fused[pass] = learner.UsesFusedCollection();
        learner.CopyCollected(&got[pass].obs, &got[pass].acts, &got[pass].rew, &got[pass].done);
        got[pass].T = learner.StepsPerIteration(); got[pass].agents = learner.NumAgents();
        resets[pass] = g_rewardResets; presteps[pass] = g_rewardPreSteps; callbacks[pass] = g_deferCallbacks; sums[pass] = g_deferCallbackSum;
    }
    CHECK(fused[0] && !fused[1]);                       // the deferred run stayed in the fused launch, the other one stepped with the host in between
    CHECK(got[0].T == steps && got[0].agents == envs * players);
    CHECK(got[0].acts == got[1].acts && got[0].done == got[1].done);
    int dones = 0; for (int32_t d : got[0].done) dones += d;
    CHECK(dones > 0);
    double worstObs = 0, worstRew = 0, sumAbs = 0;
    for (size_t i = 0; i < got[0].obs.size(); i++) worstObs = std::max(worstObs, (double)std::fabs(got[0].obs[i] - got[1].obs[i]));
    for (size_t i = 0; i < got[0].rew.size(); i++) { worstRew = std::max(worstRew, (double)std::fabs(got[0].rew[i] - got[1].rew[i])); sumAbs += std::fabs(got[0].rew[i]); }
    std::printf("user reward only, fused (%dv%d): %d dones, max |obs diff| %.3g, max |reward diff| %.3g (mean |reward| %.3g); Reset hooks %d / %d, PreStep %d / %d, callbacks %d / %d\n",
                teamSize, teamSize, dones / players, worstObs, worstRew, sumAbs / got[0].rew.size(), resets[0], resets[1], presteps[0], presteps[1], callbacks[0], callbacks[1]);
    CHECK(worstObs == 0.0 && worstRew <= 1e-6);
    CHECK(resets[0] == resets[1] && presteps[0] == presteps[1] && presteps[0] == 2 * envs * steps);
    CHECK(callbacks[0] == callbacks[1] && callbacks[0] == 2 * envs * steps && sums[0] == sums[1]);
    // ... and free-running (the default): game e's rows are the lockstep run's first steps[e] rows, rewards included
    {
        g_rewardResets = 0; g_rewardPreSteps = 0;
        LearnerConfig cfg = SmallConfig(envs, steps, players); cfg.lockstepCollection = false;
        Learner learner(MakeUserRewardEnv, cfg);
        learner.CollectTimesteps();
        if (learner.UsesFreeRunningCollection()) {
            Collected fr; learner.CopyCollected(&fr.obs, &fr.acts, &fr.rew, &fr.done);
            const std::vector<int32_t> st = learner.CollectedSteps(); const int N = learner.NumAgents(); long long rows = 0;
            // (first iteration of both runs: same seeds, same states)
            LearnerConfig cfg2 = SmallConfig(envs, learner.StepCapacity(), players);
            Learner lock(MakeUserRewardEnv, cfg2); lock.CollectTimesteps();
            Collected lk; lock.CopyCollected(&lk.obs, &lk.acts, &lk.rew, &lk.done);
            for (int e = 0; e < envs; e++)
                for (int t = 0; t < st[e]; t++)
                    for (int k = 0; k < players; k++) {
                        const size_t i = (size_t)t * N + (size_t)e * players + k;
                        CHECK(fr.acts[i] == lk.acts[i] && fr.done[i] == lk.done[i] && std::fabs(fr.rew[i] - lk.rew[i]) <= 1e-6f);
                        rows++;
                    }
            CHECK(rows == (long long)learner.LastIterationTimesteps());
            std::printf("user reward only, fused + free-running: %lld rows equal the lockstep run's\n", rows);
        }
    }
    return 0;
}

// ---- part 2b: the skill tracker with a custom OBSBuilder ---------------------------------------------------------
// SkillTracker.cpp:165-257 plays its eval games through GameInst::Step like any other game: whatever plugins the env has, they run.  Until
// round 5 the tracker here switched itself off when the eval match had an obs builder / terminal condition / action parser without a device
// form; now its batch steps through the same host path as the training batch (HostEnvPath.h).  Every eval episode starts with the ball a few
// ticks from one of the goal lines, so goals are scored whatever the policies do: the ratings must move, in both directions.
static std::atomic<int> g_mouthResets{0}, g_evalCallbacks{0}, g_evalGoals{0}, g_evalObsErrors{0};
struct GoalMouthSetter : StateSetter {
    GameState ResetState(Arena* arena) override {
        arena->ResetToRandomKickoff();
        const float side = (g_mouthResets++ % 2) ? 1.f : -1.f;
        BallState bs;
        bs.pos = Vec(0, side * 4900.f, 200.f); bs.vel = Vec(0, side * 2500.f, 0);
        arena->ball->SetState(bs);
        return GameState(arena);
    }
};
static EnvCreateResult MakeEvalEnv() {
    Match* match = new Match(new BallHeightReward(), {new StepLimitCondition(5)}, new RangeOBS(), new EightWayParser(), new GoalMouthSetter(), 1, true);
    return {match, new Gym(match, 8)};
}
static void EvalCallback(GameInst* game, const Gym::StepResult& r, Report&) {
    if (!game->isEval) return;
    g_evalCallbacks++;
    if (RLGSC::Math::IsBallScored(r.state.ball.pos)) g_evalGoals++;
    for (float v : r.reward) if (v != 0.f) g_evalObsErrors++;        // the eval games' reward is the zero reward (SkillTracker.cpp:10-16,51)
}
static int SkillTrackerWithUserPlugins() {
    const int envs = 8, steps = 10;
    LearnerConfig cfg = SmallConfig(envs, steps, 2);
    cfg.skillTrackerConfig.enabled = true;
    cfg.skillTrackerConfig.envCreateFunc = MakeEvalEnv;
    cfg.skillTrackerConfig.stepCallback = EvalCallback;
    cfg.skillTrackerConfig.numEnvs = 4; cfg.skillTrackerConfig.simTime = 16.f;        // 4 s per game = 60 steps of 8 ticks = 12 episodes of 5 steps
    cfg.skillTrackerConfig.updateInterval = 1; cfg.skillTrackerConfig.timestepsPerVersion = 2 * cfg.timestepsPerIteration;
    cfg.skillTrackerConfig.kickoffStatesOnly = false;                                 // (the user's setter is the point)
    cfg.skillTrackerConfig.perModeRatings = false; cfg.skillTrackerConfig.loadOldVersionsFromCheckpoints = false;
    Learner learner(MakeUserEnv, cfg);
    CHECK(learner.obsSize == 89 + 3 && learner.actionAmount == 8);
    int iterations = 0; std::vector<float> ratings;
    learner.iterationCallback = [&](Learner* l, Report& report) {
        iterations++;
        if (report.Has("Skill Rating")) ratings.push_back(report["Skill Rating"]);
        if (iterations == 4) l->config.timestepLimit = 1;
    };
    learner.Learn();
    CHECK(iterations == 4 && (int)ratings.size() == 4);
    bool moved = false; for (float r : ratings) moved = moved || r != cfg.skillTrackerConfig.initialRating;
    std::printf("skill tracker with a custom OBSBuilder / terminal condition / parser / setter: ratings %.2f %.2f %.2f %.2f; %d eval steps, %d with the ball in a goal, %d setter resets\n",
                ratings[0], ratings[1], ratings[2], ratings[3], (int)g_evalCallbacks, (int)g_evalGoals, (int)g_mouthResets);
    CHECK(moved);                                       // goals were scored and counted
    CHECK(g_evalCallbacks >= 3 * 4 * 60 && g_evalGoals > 20 && g_evalObsErrors == 0);
    CHECK(g_mouthResets >= 3 * 4 * 11);                 // the user's setter started every eval episode (5-step episodes: the user's terminal condition ran too)
    return 0;
}

// throughput of the paths that leave the device every step, for DESIGN.md: agent-steps per second of CollectTimesteps()
static void StepCounter(GameInst*, const Gym::StepResult&, Report&) {}
static int Throughput() {
    const int envs = 1024, steps = 16;
    for (int mode = 0; mode < 3; mode++) {
        g_host = mode == 2; g_hostParser = false; g_teamSize = 1;
        Learner learner(MakeBuiltinEnv, SmallConfig(envs, steps, 2));
        if (mode == 1) learner.stepCallback = StepCounter;
        learner.CollectTimesteps();
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 3; i++) learner.CollectTimesteps();
        std::vector<int32_t> d; learner.CopyCollected(nullptr, nullptr, nullptr, &d);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const char* name[3] = {"all plugins on the device", "device plugins + step callback", "reward, obs, terminal conditions on the host"};
        std::printf("collection, %d envs 1v1: %-46s %10.0f agent-steps/s\n", envs, name[mode], 3.0 * envs * 2 * steps / secs);
    }
    return 0;
}

// collectionDuringLearn in the C++ host: the epochs of iteration k on their own stream beside the collection of k + 1
static int CollectionDuringLearn() {
    g_host = false; g_hostParser = false; g_teamSize = 1; g_spawnOpponents = true;
    LearnerConfig cfg = SmallConfig(256, 16, 2);
    cfg.collectionDuringLearn = true; cfg.ppo.epochs = 2; cfg.ppo.miniBatchSize = cfg.ppo.batchSize / 2;
    cfg.ppo.autocastLearn = true;   // the overlap needs standalone inference (the fused inference kernel: bf16 mode); in fp32 mode the Learner keeps collection and learning in sequence
    Learner learner(MakeBuiltinEnv, cfg);
    int iterations = 0, withStats = 0; bool finite = true;
    learner.iterationCallback = [&](Learner* l, Report& report) {
        iterations++;
        if (report.Has("Policy Entropy")) {   // the epochs of the PREVIOUS iteration: absent from the first report
            withStats++;
            finite = finite && std::isfinite(report["Policy Entropy"]) && report["Policy Entropy"] > 0 && std::isfinite(report["Value Function Loss"]);
        }
        if (iterations == 4) l->config.timestepLimit = 1;
    };
    learner.Learn();
    CHECK(iterations == 4 && withStats == 3 && finite);
    CHECK(learner.totalTimesteps == 4ull * 256 * 2 * 16 && learner.totalEpochs == 8);
    std::printf("collectionDuringLearn: 4 iterations, PPO statistics in reports 2..4\n");
    {   // fp32 mode: inference would share the learner's activation scratch with the epochs on the other stream -> serial, statistics in every report
        LearnerConfig c2 = SmallConfig(64, 8, 2);
        c2.collectionDuringLearn = true; c2.ppo.autocastLearn = false;
        Learner l2(MakeBuiltinEnv, c2);
        int its = 0, stats = 0;
        l2.iterationCallback = [&](Learner* l, Report& report) { its++; if (report.Has("Policy Entropy")) stats++; if (its == 2) l->config.timestepLimit = 1; };
        l2.Learn();
        CHECK(its == 2 && stats == 2);
        std::printf("collectionDuringLearn in fp32 mode: falls back to serial (statistics in every report)\n");
    }
    return 0;
}

// ---- part 3: the standalone Gym ------------------------------------------------------------------------------------------------------
static int StandaloneGym() {
    RocketSim::Math::SeedRandEngine(5);
    CombinedReward reward({{new VelocityPlayerToBallReward(), 1.f}, {new EventReward({.touch = 1.f}), 1.f}}, true);
    NoTouchCondition noTouch(400); GoalScoreCondition goal; DefaultOBS obs; DiscreteAction parser; KickoffState kickoff;
    Match match(&reward, {&noTouch, &goal}, &obs, &parser, &kickoff, 1, true);
    Gym gym(&match, 8);
    CHECK(gym.arena && gym.arena->_cars.size() == 2 && gym.carIds.size() == 2 && gym.arena->_boostPads.size() == 34 && gym.actionDelay == 7);
    FList2 first = gym.Reset();
    CHECK(first.size() == 2 && first[0].size() == 89);
    const CarState blue = gym.arena->_cars[0]->GetState(), orange = gym.arena->_cars[1]->GetState();
    CHECK(blue.pos.y < 0 && std::fabs(blue.pos.x + orange.pos.x) < 1e-3f && std::fabs(blue.pos.y + orange.pos.y) < 1e-3f && blue.pos.z == 17.f);   // mirrored kickoff spots
    CHECK(gym.arena->ball->GetState().pos.z == RLConst::BALL_REST_Z && gym.prevState.players.size() == 2);
    // the arena can be edited through the facade between calls: line both cars up with the ball
    for (Car* car : gym.arena->_cars) {
        CarState cs = car->GetState();
        const float side = car->team == Team::BLUE ? -1.f : 1.f;
        cs.pos = Vec(0, 2000 * side, 17); cs.rotMat = Angle(-side * 1.5707963f, 0, 0).ToRotMat();
        car->SetState(cs);
    }
    // full throttle + boost straight at the ball (action row: throttle 1, steer 0, ..., boost 1): both cars reach it
    ActionParser::Input drive(2, 0);
    {
        ActionSet probe; int best = -1;
        for (int a = 0; a < parser.GetActionAmount(); a++) {
            probe = parser.ParseActions({a, a}, gym.prevState);
            if (probe[0].throttle == 1 && probe[0].steer == 0 && probe[0].boost == 1 && probe[0].jump == 0 && probe[0].handbrake == 0 && probe[0].pitch == 0 && probe[0].yaw == 0) { best = a; break; }
        }
        CHECK(best >= 0);
        drive = {best, best};
    }
    const uint64_t tick0 = gym.arena->tickCount;
    bool touched = false, done = false; float total = 0; int stepsTaken = 0;
    for (int i = 0; i < 40 && !done; i++) {
        Gym::StepResult r = gym.Step(drive);
        stepsTaken++;
        CHECK(r.obs.size() == 2 && r.obs[0].size() == 89 && r.reward.size() == 2 && r.state.players.size() == 2);
        CHECK(r.state.deltaTickCount == (i == 0 ? 1 : 8));                                                  // GameState is taken one tick into the step
        CHECK(gym.arena->tickCount == tick0 + 8ull * (uint64_t)(i + 1) && r.state.lastTickCount == gym.arena->tickCount - 7);
        touched = touched || r.state.players[0].ballTouchedStep || r.state.players[1].ballTouchedStep;
        total += r.reward[0];
        done = r.done;
    }
    CHECK(touched && total > 1.f && gym.totalSteps == stepsTaken && gym.totalTicks == 8 * stepsTaken);
    // the arena can be edited between steps through the facade: teleport the ball into the orange goal -> GoalScoreCondition ends the episode
    BallState in; in.pos = Vec(0, 5300, 200); in.vel = Vec(0, 500, 0);
    gym.arena->ball->SetState(in);
    Gym::StepResult scored = gym.Step(drive);
    CHECK(scored.done && scored.state.scoreLine[0] == 1 && scored.state.scoreLine[1] == 0);
    // Arena::Step on its own (no gym layer): the ball falls
    Arena* arena = Arena::Create(GameMode::SOCCAR);
    arena->AddCar(Team::BLUE); arena->AddCar(Team::ORANGE);
    arena->ResetToRandomKickoff(123);
    BallState up; up.pos = Vec(0, 0, 1000); up.vel = Vec(0, 0, -1);   // (a ball at rest sleeps until something touches it, as in the reference)
    arena->ball->SetState(up);
    arena->Step(60);
    const BallState fell = arena->ball->GetState();
    CHECK(arena->tickCount == 60 && fell.pos.z < 1000.f - 70.f && fell.pos.z > 1000.f - 90.f && fell.vel.z < -300.f);   // 0.5 s of -650 uu/s^2
    delete arena;
    {   // goal / bump callbacks of the arena facade (Arena.cpp:335-413, 804-808), raised from the state every tick leaves behind
        Arena* cb = Arena::Create(GameMode::SOCCAR);
        Car* blueCar = cb->AddCar(Team::BLUE); Car* orangeCar = cb->AddCar(Team::ORANGE);
        cb->ResetToRandomKickoff(7);
        struct Seen { int goals = 0, bumps = 0, demos = 0; Team scorer = Team::ORANGE; Car *bumper = nullptr, *victim = nullptr; uint64_t bumpTick = 0; } seen;
        cb->SetGoalScoreCallback([](Arena*, Team t, void* u) { Seen* s = (Seen*)u; s->goals++; s->scorer = t; }, &seen);
        cb->SetCarBumpCallback([](Arena* a, Car* b, Car* v, bool demo, void* u) { Seen* s = (Seen*)u; s->bumps++; s->demos += demo ? 1 : 0; s->bumper = b; s->victim = v; s->bumpTick = a->tickCount; }, &seen);
        // a supersonic blue car drives into a standing orange one: one demo, raised once (the cooldown keeps the pair quiet afterwards)
        CarState a = blueCar->GetState(); a.pos = Vec(0, -400, 17); a.rotMat = Angle(1.5707963f, 0, 0).ToRotMat(); a.vel = Vec(0, 2300, 0); a.isSupersonic = true; a.boost = 100; blueCar->SetState(a);
        CarState o = orangeCar->GetState(); o.pos = Vec(0, 0, 17); o.rotMat = Angle(0, 0, 0).ToRotMat(); o.vel = Vec(0, 0, 0); orangeCar->SetState(o);
        blueCar->controls.throttle = 1; blueCar->controls.boost = true;
        BallState away; away.pos = Vec(3000, 3000, 93.15f); cb->ball->SetState(away);
        cb->Step(40);
        CHECK(seen.bumps == 1 && seen.demos == 1 && seen.bumper == blueCar && seen.victim == orangeCar && seen.bumpTick > 0 && seen.bumpTick < 40);
        CHECK(orangeCar->GetState().isDemoed && seen.goals == 0);
        // the ball crosses the orange goal line: blue scores, and the callback fires on every tick that ends with the ball behind the line
        BallState in2; in2.pos = Vec(0, 5100, 300); in2.vel = Vec(0, 1500, 0); cb->ball->SetState(in2);
        cb->Step(12);
        CHECK(cb->IsBallScored() && seen.goals >= 1 && seen.goals < 12 && seen.scorer == Team::BLUE);
        const int g0 = seen.goals;
        cb->Step(3);
        CHECK(seen.goals == g0 + 3);
        delete cb;
    }
    {   // MutatorConfig through the facade (Arena::SetMutatorConfig, Gym's constructor argument): the run-time fields reach the device, the compiled-in ones are refused
        Arena* mu = Arena::Create(GameMode::SOCCAR);
        Car* car = mu->AddCar(Team::BLUE); mu->AddCar(Team::ORANGE);
        mu->ResetToRandomKickoff(5);
        MutatorConfig mc(GameMode::SOCCAR);
        mc.gravity = Vec(0, 0, -325.f); mc.ballMaxSpeed = 1000.f; mc.carSpawnBoostAmount = 77.f; mc.demoMode = DemoMode::DISABLED; mc.boostUsedPerSecond = 0.f;
        mu->SetMutatorConfig(mc);
        BallState up; up.pos = Vec(0, 0, 1500); up.vel = Vec(0, 0, -1); mu->ball->SetState(up);
        car->controls.throttle = 1; car->controls.boost = true;
        const float boost0 = car->GetState().boost;
        mu->Step(60);
        const BallState fell = mu->ball->GetState();
        CHECK(fell.pos.z < 1500.f - 35.f && fell.pos.z > 1500.f - 45.f && fell.vel.z < -150.f && fell.vel.z > -170.f);   // 0.5 s of -325 uu/s^2, half of what the default drops it
        CHECK(car->GetState().boost == boost0);                                                                          // boosting costs nothing under boostUsedPerSecond = 0
        BallState fast; fast.pos = Vec(0, 0, 1000); fast.vel = Vec(3000, 0, 0); mu->ball->SetState(fast);
        mu->Step(2);
        CHECK(mu->ball->GetState().vel.Length() <= 1000.5f);                                                             // ballMaxSpeed
        CHECK(mu->GetMutatorConfig().gravity.z == -325.f && !mu->GetMutatorConfig().IsDefault());
        bool refused = false;
        MutatorConfig big = mc; big.ballRadius = 120.f;
        try { mu->SetMutatorConfig(big); } catch (const std::exception&) { refused = true; }
        CHECK(refused);
        refused = false;
        MutatorConfig side = mc; side.carMass = 200.f;
        try { mu->SetMutatorConfig(side); } catch (const std::exception&) { refused = true; }
        CHECK(refused);
        delete mu;
        // the same through Gym's constructor (SIM/Gym.cpp:40-44)
        CombinedReward r0({{new VelocityPlayerToBallReward(), 1.f}}, true);
        NoTouchCondition nt0(50);
        Match mm(&r0, {&nt0}, &obs, &parser, &kickoff, 1, true);
        Gym gm(&mm, 8, CAR_CONFIG_OCTANE, GameMode::SOCCAR, mc);
        gm.Reset();
        BallState up2; up2.pos = Vec(0, 0, 1500); up2.vel = Vec(0, 0, -1); gm.arena->ball->SetState(up2);
        for (int i = 0; i < 8; i++) gm.Step({drive[0], drive[0]});                                                       // 64 ticks
        const float z = gm.arena->ball->GetState().pos.z;
        CHECK(z < 1500.f - 40.f && z > 1500.f - 52.f);
    }
    {   // a gym without opponents: one car, one observation row with no other player in it
        CombinedReward solo({{new VelocityPlayerToBallReward(), 1.f}}, true);
        NoTouchCondition nt(4);
        Match m1(&solo, {&nt}, &obs, &parser, &kickoff, 1, false);
        Gym g1(&m1, 8);
        FList2 o1 = g1.Reset();
        CHECK(m1.playerAmount == 1 && g1.arena->_cars.size() == 1 && o1.size() == 1 && o1[0].size() == 51 + 19 && g1.prevState.players.size() == 1 && g1.prevState.players[0].carId == 1);
        bool ended = false;
        for (int i = 0; i < 4; i++) { Gym::StepResult r = g1.Step({drive[0]}); CHECK(r.obs.size() == 1 && r.reward.size() == 1 && r.state.players.size() == 1); ended = r.done; }
        CHECK(ended);   // NoTouchCondition(4)
    }
    std::printf("standalone gym: %d steps to the first touch-and-beyond, reward %.2f\n", stepsTaken, total);
    return 0;
}

// Free-running collection (LearnerConfig::lockstepCollection = false, the default; ThreadAgentManager.cpp:16-82): every game steps at its own
// pace until the batch has timestepsPerIteration together.  What a game's players experienced does not depend on the pace: its rows are the first
// rows of a lockstep learner's with the same seed.  Then the iteration goes through AddNewExperience / LearnPPO with its ragged trajectories.
static int FreeRunning() {
    const int envs = 256, S = 12;
    g_host = false; g_hostParser = false; g_teamSize = 1; g_spawnOpponents = true;
    LearnerConfig lock = SmallConfig(envs, 2 * S, 2);
    LearnerConfig free = SmallConfig(envs, S, 2); free.lockstepCollection = false;
    Learner a(MakeBuiltinEnv, lock), b(MakeBuiltinEnv, free);
    CHECK(b.StepCapacity() == 2 * S && a.StepsPerIteration() == 2 * S && b.StepsPerIteration() == S);
    a.CollectTimesteps(); b.CollectTimesteps();
    CHECK(b.UsesFreeRunningCollection() && !a.UsesFreeRunningCollection());
    Collected ca, cb;
    a.CopyCollected(&ca.obs, &ca.acts, &ca.rew, &ca.done); b.CopyCollected(&cb.obs, &cb.acts, &cb.rew, &cb.done);
    const std::vector<int32_t> st = b.CollectedSteps();
    const int N = b.NumAgents(), D = b.obsSize, P = 2;
    int64_t rows = 0; int lo = 1 << 30, hi = 0;
    for (int e = 0; e < envs; e++) { rows += (int64_t)st[e] * P; lo = std::min(lo, (int)st[e]); hi = std::max(hi, (int)st[e]); }
    CHECK(rows == (int64_t)b.LastIterationTimesteps() && rows >= (int64_t)envs * P * S && rows < (int64_t)envs * P * (S + 1) + 1 && hi <= 2 * S);
    size_t bad = 0;
    for (int e = 0; e < envs; e++)
        for (int t = 0; t < st[e]; t++)
            for (int k = 0; k < P; k++) {
                const size_t i = (size_t)t * N + (size_t)e * P + k;
                if (ca.acts[i] != cb.acts[i] || ca.rew[i] != cb.rew[i] || ca.done[i] != cb.done[i]) bad++;
                if (std::memcmp(&ca.obs[((size_t)(t + 1) * N + (size_t)e * P + k) * D], &cb.obs[((size_t)(t + 1) * N + (size_t)e * P + k) * D], (size_t)D * 4)) bad++;
            }
    std::printf("free-running collection: %d games made %d..%d steps (%lld agent-steps for a batch of %d), rows differing from the lockstep learner's: %zu\n", envs, lo, hi, (long long)rows, envs * P * S, bad);
    CHECK(bad == 0);
    // three whole iterations: the second starts from every game's OWN last observation
    for (int it = 0; it < 3; it++) {
        Report rep;
        if (it) b.CollectTimesteps();
        b.AddNewExperience(rep); b.LearnPPO(rep);
        CHECK(rep.Has("Policy Entropy") && std::isfinite(rep["Policy Entropy"]) && std::isfinite(rep["Avg Return"]) && std::isfinite(rep["Value Function Loss"]));
        CHECK((int64_t)b.LastIterationTimesteps() >= (int64_t)envs * P * S);
    }
    // ... and what the second launch read as its first observations is what the first wrote after each game's last step
    {
        Learner c(MakeBuiltinEnv, free);
        c.CollectTimesteps();
        Collected c1; c.CopyCollected(&c1.obs, nullptr, nullptr, nullptr);
        const std::vector<int32_t> s1 = c.CollectedSteps();
        c.CollectTimesteps();
        Collected c2; c.CopyCollected(&c2.obs, nullptr, nullptr, nullptr);
        size_t off = 0;
        for (int e = 0; e < envs; e++)
            for (int k = 0; k < P; k++)
                if (std::memcmp(&c1.obs[((size_t)s1[e] * N + (size_t)e * P + k) * D], &c2.obs[((size_t)e * P + k) * D], (size_t)D * 4)) off++;
        CHECK(off == 0);
    }
    return 0;
}

int main(int argc, char** argv) {
    RocketSim::Init("./collision_meshes", true);
    setenv("RLGPU_QUIET", "1", 1);
    try {
        if (argc > 1 && std::string(argv[1]) == "throughput") return Throughput();
        if (ComparePaths(1, false)) return 1;
        if (ComparePaths(2, true)) return 1;
        if (ComparePaths(2, false, false)) return 1;    // Match(..., spawnOpponents = false): two blue cars and nobody else
        if (ComparePaths(1, true, false)) return 1;     // a single car
        if (UserPlugins()) return 1;
        if (DeferredUserReward(1)) return 1;
        if (DeferredUserReward(2)) return 1;
        if (SkillTrackerWithUserPlugins()) return 1;
        if (CollectionDuringLearn()) return 1;
        if (FreeRunning()) return 1;
        if (StandaloneGym()) return 1;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 1;
    }
    std::printf("plugin fallback ok\n");
    return 0;
}
