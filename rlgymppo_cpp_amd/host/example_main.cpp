// example_main.cpp -- a training program written against the reference's public API (compare /examplemain.cpp of RLGymPPO_CPP:
// same headers, same classes, same config fields), built against this repo's headers and librlgymppo_amd.so instead.
//   usage: example_main [iterations] [numThreads] [numGamesPerThread] [timestepsPerIteration] [checkpoint folder] [render]
#include <RLGymPPO_CPP/Learner.h>

#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CombinedReward.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/StateSetters/RandomState.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>

using namespace RLGPC;
using namespace RLGSC;

static int g_iterations_left = 2;

// per-step metrics the reference's way (examplemain.cpp:23-36).  A step callback makes the learner hand every game's GameState to the
// host each step; LearnerConfig::deviceStepMetrics gives the same three numbers from the step kernels -- this program uses that unless
// EXAMPLE_STEP_CALLBACK is set
static void StepMetrics(GameInst* game, const Gym::StepResult& result, Report& metrics) {
    for (const PlayerData& player : result.state.players) {
        metrics.AccumAvg("player_speed", player.phys.vel.Length());
        metrics.AccumAvg("ball_touch_ratio", player.ballTouchedStep);
        metrics.AccumAvg("in_air_ratio", !player.carState.isOnGround);
    }
}

static void IterationMetrics(Learner* learner, Report& all) {
    AvgTracker speed, touch, air;
    for (const Report& r : learner->GetAllGameMetrics()) {
        if (!r.Has("player_speed_avg_count")) continue;
        speed += (float)r.GetAvg("player_speed"); touch += (float)r.GetAvg("ball_touch_ratio"); air += (float)r.GetAvg("in_air_ratio");
    }
    if (speed.count > 0) { all["player_speed"] = speed.Get(); all["ball_touch_ratio"] = touch.Get(); all["in_air_ratio"] = air.Get(); }
    if (--g_iterations_left <= 0 && !getenv("EXAMPLE_TIMESTEP_LIMIT")) learner->config.timestepLimit = 1;   // stop after this iteration
}

static EnvCreateResult MakeEnv() {
    const int tickSkip = 8;
    const float noTouchSeconds = 10.f;
    auto* reward = new CombinedReward({
        {new FaceBallReward(), 0.1f},
        {new VelocityPlayerToBallReward(), 0.5f},
        {new VelocityBallToGoalReward(), 1.0f},
        {new EventReward({.teamGoal = 1.f, .concede = -1.f}), 50.f},
    });
    std::vector<TerminalCondition*> terminal = {new NoTouchCondition((int)(noTouchSeconds * 120 / tickSkip)), new GoalScoreCondition()};
    Match* match = new Match(reward, terminal, new DefaultOBS(), new DiscreteAction(), new RandomState(true, true, true), 1, true);
    return {match, new Gym(match, tickSkip)};
}

int main(int argc, char** argv) {
    g_iterations_left = argc > 1 ? atoi(argv[1]) : 2;
    RocketSim::Init("./collision_meshes");

    LearnerConfig cfg = {};
    cfg.numThreads = argc > 2 ? atoi(argv[2]) : 16;
    cfg.numGamesPerThread = argc > 3 ? atoi(argv[3]) : 24;
    int tsPerItr = argc > 4 ? atoi(argv[4]) : 100 * 1000;
    cfg.timestepsPerIteration = tsPerItr;
    cfg.ppo.batchSize = tsPerItr;
    cfg.ppo.miniBatchSize = 25 * 1000;
    cfg.expBufferSize = tsPerItr * 3;
    cfg.ppo.epochs = 1;
    cfg.ppo.entCoef = 0.01f;
    cfg.ppo.policyLR = 2e-4f;
    cfg.ppo.criticLR = 2e-4f;
    cfg.ppo.policyLayerSizes = {256, 256, 256};
    cfg.ppo.criticLayerSizes = {256, 256, 256};
    cfg.ppo.autocastLearn = true;
    cfg.sendMetrics = true;    // JSON lines under ./metrics/<project>/<run id>.jsonl
    cfg.renderMode = argc > 6 && std::string(argv[6]) == "render";   // play the newest checkpoint for RocketSimVis instead of training
    if (getenv("EXAMPLE_SKILL_TRACKER")) {   // ELO evaluation against stored versions of the policy, a new version every iteration
        cfg.skillTrackerConfig.enabled = true; cfg.skillTrackerConfig.numEnvs = 4; cfg.skillTrackerConfig.simTime = 16; cfg.skillTrackerConfig.updateInterval = 1;
        cfg.skillTrackerConfig.timestepsPerVersion = tsPerItr; cfg.skillTrackerConfig.maxVersions = 2;
        cfg.timestepsPerSave = tsPerItr;   // every version has its checkpoint, so a resumed run finds them (Learner.cpp:311-370)
    }
    const bool useStepCallback = getenv("EXAMPLE_STEP_CALLBACK") != nullptr;
    cfg.deviceStepMetrics = !useStepCallback;
    cfg.checkpointSaveFolder = argc > 5 ? argv[5] : "";
    cfg.checkpointLoadFolder = cfg.checkpointSaveFolder;

    try {
        Learner learner(MakeEnv, cfg);
        if (cfg.renderMode) learner.config.timestepLimit = learner.totalTimesteps + (uint64_t)std::max(g_iterations_left, 1) * 2;   // here "iterations" = rendered 1v1 steps
        if (useStepCallback) learner.stepCallback = StepMetrics;
        learner.iterationCallback = IterationMetrics;
        if (const char* lim = getenv("EXAMPLE_TIMESTEP_LIMIT")) learner.config.timestepLimit = strtoull(lim, nullptr, 10);   // run to a timestep count instead of an iteration count
        learner.Learn();
        std::cerr << "[example_main rank " << learner.Rank() << "/" << learner.WorldSize() << "] " << learner.totalIterations << " iterations, " << learner.totalTimesteps << " timesteps" << std::endl;
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 1;
    }
    return 0;
}
