// bench_main.cpp -- the measured process of bench.py: the reference's program shape (examplemain.cpp: EnvCreateFn + LearnerConfig +
// Learner) on include/RLGymPPO_CPP over librlgymppo_amd.so / librlgpu.so.  No Python, no torch: the HIP runtime, the C-ABI library
// and (multi-GPU) RCCL are all that touch the device.  Prints ONE JSON object on stdout (rank 0).
//   bench_main --envs E --team-size S --horizon T --steps K --warmup W [--epochs n] [--padded-zero-sum] [--fp32 | --fp16] [--overlap] [--trained-warmup I --trained-steps J]
//              [--learned-warmup I --learned-epochs n --learned-steps J] [--mesh-dir DIR]
// One "step" = one full PPO iteration: T gym steps of every env with on-device policy inference, value pass + GAE, shuffled
// minibatches (4 per batch), clip + Adam.  With --trained-warmup the same measurement is repeated after I more iterations, when
// the policy has started to play and contacts are more frequent ("trained_regime").  With --learned-warmup the run then goes on LEARNING
// (--learned-epochs PPO epochs per iteration, what tools/train_probe.py uses) for I more iterations -- by then the policy chases and hits the
// ball -- and is measured once more with the headline's settings ("trained_regime_learned"; mean step reward and entropy are printed with it).
// --mesh-dir: a directory of .cmf collision meshes for RocketSim::Init (default ./collision_meshes; absent -> the procedural arena).
#include <RLGymPPO_CPP/Learner.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CombinedReward.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/ZeroSumReward.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBSPadded.h>
#include <RLGymSim_CPP/Utils/StateSetters/RandomState.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>
#include <hip/hip_runtime_api.h>
#include "../../include/rlgpu.h"
#include <algorithm>
#include <chrono>
#include <vector>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

using namespace RLGPC;
using namespace RLGSC;
extern char** environ;

static int g_team = 1; static bool g_padded = false, g_userReward = false;
// --user-reward: what every RLGym user writes -- a reward class of their own (here a subclass of a built-in that changes nothing: the Learner cannot know
// that, a subclass has no device form).  The reward then runs on the host; with LearnerConfig::deferHostRewards it does so after the fused collection launch
struct MyVelocityPlayerToBallReward : VelocityPlayerToBallReward {};

static EnvCreateResult EnvCreateFunc() {   // examplemain.cpp:58-100
    constexpr int TICK_SKIP = 8; constexpr float NO_TOUCH_TIMEOUT_SECS = 10.f;
    auto rewards = new CombinedReward({ { new FaceBallReward(), 0.1f }, { g_userReward ? (RewardFunction*)new MyVelocityPlayerToBallReward() : new VelocityPlayerToBallReward(), 0.5f }, { new VelocityBallToGoalReward(), 1.0f },
        { new EventReward({ .teamGoal = 1.f, .concede = -1.f }), 50.f } });
    RewardFunction* root = rewards;
    if (g_padded) root = new ZeroSumReward(rewards, 0.3f, 1.0f);
    std::vector<TerminalCondition*> terminalConditions = { new NoTouchCondition(NO_TOUCH_TIMEOUT_SECS * 120 / TICK_SKIP), new GoalScoreCondition() };
    OBSBuilder* obs = g_padded ? (OBSBuilder*)new DefaultOBSPadded(g_team) : (OBSBuilder*)new DefaultOBS();
    auto match = new Match(root, terminalConditions, obs, new DiscreteAction(), new RandomState(true, true, true), g_team, true);
    Gym* gym = new Gym(match, TICK_SKIP);
    return { match, gym };
}

struct Timed { double stepReward = 0, entropy = 0; double agentSteps = 0; double sec; float envMs; int envLaunches; float gemmMs; double gemmFlops; int gemmCalls; double consumeMs; std::vector<double> rankSec; float arMs; int arCalls; };

int main(int argc, char* argv[]) {
    int envs = 4096, horizon = 32, steps = 200, warmup = 20, epochs = 1, trainedWarm = 0, trainedSteps = 0, learnedWarm = 0, learnedEpochs = 2, learnedSteps = 0, collectQueue = -1; bool fp32 = false, fp16 = false, overlap = false, lockstep = false, deterministic = false, noDefer = false;
    std::string meshDir = "./collision_meshes", saveDir, loadDir;   // --save-checkpoint DIR: Learner::Save() at the end; --load-checkpoint DIR: start from its newest checkpoint (a policy that has learned, on another mesh)
    for (int i = 1; i < argc; i++) {
        auto is = [&](const char* k) { return !strcmp(argv[i], k); };
        if (is("--envs")) envs = atoi(argv[++i]); else if (is("--team-size")) g_team = atoi(argv[++i]); else if (is("--horizon")) horizon = atoi(argv[++i]);
        else if (is("--steps")) steps = atoi(argv[++i]); else if (is("--warmup")) warmup = atoi(argv[++i]); else if (is("--epochs")) epochs = atoi(argv[++i]);
        else if (is("--padded-zero-sum")) g_padded = true; else if (is("--fp32")) fp32 = true; else if (is("--overlap")) overlap = true; else if (is("--fp16")) fp16 = true; else if (is("--lockstep")) lockstep = true; else if (is("--deterministic")) deterministic = true; else if (is("--collect-queue")) collectQueue = atoi(argv[++i]);
        else if (is("--trained-warmup")) trainedWarm = atoi(argv[++i]); else if (is("--trained-steps")) trainedSteps = atoi(argv[++i]);
        else if (is("--learned-warmup")) learnedWarm = atoi(argv[++i]); else if (is("--learned-epochs")) learnedEpochs = atoi(argv[++i]); else if (is("--learned-steps")) learnedSteps = atoi(argv[++i]);
        else if (is("--mesh-dir")) meshDir = argv[++i];
        else if (is("--user-reward")) g_userReward = true; else if (is("--no-defer")) noDefer = true;
        else if (is("--save-checkpoint")) saveDir = argv[++i]; else if (is("--load-checkpoint")) loadDir = argv[++i];
        else { fprintf(stderr, "bench_main: unknown argument %s\n", argv[i]); return 2; }
    }
    // every RLGPU_* variable this process was started with goes into the JSON line ("env_overrides"): a number measured under a path selector or on the
    // test transport says so itself (the launcher's rendezvous plumbing -- directory, tag, timeouts -- is left out)
    std::string overrides;
    for (char** e = environ; e && *e; e++) {
        if (strncmp(*e, "RLGPU_", 6) != 0) continue;
        std::string kv(*e); const std::string name = kv.substr(0, kv.find('='));
        if (name == "RLGPU_QUIET" || name == "RLGPU_COMM_DIR" || name == "RLGPU_COMM_TAG" || name == "RLGPU_COMM_TIMEOUT_S" || name == "RLGPU_COMM_STALE_S") continue;
        for (char& ch : kv) if (ch == '"' || ch == '\\') ch = '?';
        overrides += std::string(overrides.empty() ? "" : ", ") + "\"" + kv + "\"";
    }
    const char* transport_env = getenv("RLGPU_COMM_TRANSPORT");
    setenv("RLGPU_QUIET", "1", 1);
    RocketSim::Init(meshDir, true);
    const int64_t nAgents = (int64_t)envs * 2 * g_team, B = nAgents * horizon;
    LearnerConfig cfg = {};
    cfg.numThreads = 1; cfg.numGamesPerThread = envs;
    if (g_userReward) {   // host plugins run on numThreads host threads (one plugin set per game, as in the reference): as many as the host has, dividing the batch
        int th = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
        while (th > 1 && envs % th != 0) th--;
        cfg.numThreads = th; cfg.numGamesPerThread = envs / th;
        cfg.deferHostRewards = !noDefer;
    }
    cfg.timestepsPerIteration = B; cfg.expBufferSize = B;
    cfg.ppo.batchSize = B; cfg.ppo.miniBatchSize = B / 4; cfg.ppo.epochs = epochs;
    cfg.ppo.policyLR = 2e-4f; cfg.ppo.criticLR = 2e-4f; cfg.ppo.entCoef = 0.01f; cfg.ppo.autocastLearn = !fp32;
    cfg.ppo.policyLayerSizes = { 256, 256, 256 }; cfg.ppo.criticLayerSizes = { 256, 256, 256 };
    cfg.randomSeed = 123; cfg.sendMetrics = false; cfg.checkpointSaveFolder = saveDir; cfg.checkpointLoadFolder = loadDir; cfg.timestepsPerSave = (int64_t)1 << 60;
    cfg.timestepLimit = 0;
    cfg.collectStepQueue = collectQueue;
    cfg.deterministicGradients = deterministic;   // fixed-order gradient sums + lockstep collection: the run is a function of its seed (tests)
    cfg.lockstepCollection = lockstep;     // default: the reference's free-running agents (every game at its own pace until the batch is full)
    if (fp16) setenv("RLGPU_AUTOCAST_FP16", "1", 1);   // fp16 operands + dynamic loss scale in the minibatch kernels (BASELINE configs[4]'s "fp16 autocast"; Learner.hip reads it at construction)
    cfg.collectionDuringLearn = overlap;   // not the headline: the reference's default pauses collection while it learns
    Learner learner(EnvCreateFunc, cfg);
    const int rank = learner.Rank(), world = learner.WorldSize();

    auto barrier = [&]() { (void)hipDeviceSynchronize(); (void)learner.MaxOverRanks(0.0); (void)hipDeviceSynchronize(); };   // device sync + a collective = a barrier
    double lastStepReward = 0, lastEntropy = 0;
    auto iteration = [&](double* consumeMs) {
        Report rep;
        struct Keep { Report& r; double& a; double& b; ~Keep() { if (r.Has("Average Step Reward")) a = r["Average Step Reward"]; if (r.Has("Policy Entropy")) b = r["Policy Entropy"]; } } keep{rep, lastStepReward, lastEntropy};
        learner.totalIterations++;
        learner.CheckReplicas();
        learner.CollectTimesteps();
        if (overlap) { learner.FinishLearn(rep); learner.AddNewExperience(rep); learner.LearnPPO(rep); return; }   // the epochs run beside the next collection
        if (consumeMs) (void)hipDeviceSynchronize();   // the collection launch is asynchronous: the consumption clock starts when it has finished
        auto t0 = std::chrono::steady_clock::now();
        learner.AddNewExperience(rep);
        learner.LearnPPO(rep);   // ends with a stream sync (it reads its metrics)
        if (consumeMs) *consumeMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    auto measure = [&](int w, int k) {
        for (int i = 0; i < w; i++) iteration(nullptr);
        barrier();
        Timed t{}; float a; int b; float c; double d; int e;
        learner.DeviceTimings(a, b, c, d, e, true);
        learner.AllReduceTimings(a, b, true);
        auto t0 = std::chrono::steady_clock::now();
        const uint64_t ts0 = learner.totalTimesteps;
        for (int i = 0; i < k; i++) iteration(&t.consumeMs);
        barrier();
        t.agentSteps = (double)(learner.totalTimesteps - ts0);   // what the iterations really gathered, all ranks (a free-running iteration holds B .. B + one step of every game)
        t.rankSec = learner.GatherOverRanks(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());   // every rank's own clock
        t.sec = *std::max_element(t.rankSec.begin(), t.rankSec.end());                                                        // the slowest rank's
        learner.AllReduceTimings(t.arMs, t.arCalls, false);
        learner.DeviceTimings(t.envMs, t.envLaunches, t.gemmMs, t.gemmFlops, t.gemmCalls, false);
        t.consumeMs /= std::max(1, k);
        t.stepReward = lastStepReward; t.entropy = lastEntropy;
        return t;
    };
    Timed m = measure(warmup, steps);
    Timed tr{}; bool haveTr = false;
    if (trainedWarm > 0 && trainedSteps > 0) { tr = measure(trainedWarm, trainedSteps); haveTr = true; }
    Timed ln{}; bool haveLn = false;
    if (learnedWarm > 0 && learnedSteps > 0) {   // learn on with more epochs per iteration, then measure with the headline's settings again
        learner.config.ppo.epochs = learnedEpochs;
        for (int i = 0; i < learnedWarm; i++) iteration(nullptr);
        learner.config.ppo.epochs = epochs;
        ln = measure(0, learnedSteps); haveLn = true;
    }
    learner.CheckReplicas();
    if (!saveDir.empty() && rank == 0) learner.Save();
    fprintf(stderr, "[bench_main rank %d/%d] %.3f s for %d iterations, parameter checksum %016llx, sampler stream %u, env seed %u\n", rank, world, m.sec, steps,
            (unsigned long long)learner.ParamChecksum(), learner.SamplerStream(), (unsigned)cfg.randomSeed + 1000u * (unsigned)rank);
    if (rank == 0) {
        const int nP = 2 * g_team, D = learner.obsSize;
        // SURVEY 8d, verbatim: algorithmic bytes per gym step per env = 2 (336 N_p + 264) + N_p (4 D + 8) + 4
        const double A = 2.0 * (336.0 * nP + 264.0) + nP * (4.0 * D + 8.0) + 4.0;
        const bool fused = learner.UsesFusedCollection();
        // gym steps of every env per collection launch, on average (free-running: what the launches really made)
        const double stepsPerLaunch = fused ? (m.envLaunches ? m.agentSteps / world / (double)nAgents / m.envLaunches : horizon) : 1;
        printf("{\"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"envs_per_gpu\": %d, \"team_size\": %d, \"horizon\": %d, \"batch\": %lld, \"minibatch\": %lld, \"epochs\": %d, \"obs_size\": %d, "
               "\"elapsed_s\": %.6f, \"agent_steps\": %.0f, \"value\": %.3f, \"ms_per_step\": %.6f, \"ppo_iter_ms\": %.6f, \"fused_collect\": %s, \"collection\": \"%s\", "
               "\"env_kernel_ms_total\": %.4f, \"env_launches\": %d, \"algorithmic_bytes_per_gym_step_per_env\": %.0f, \"gym_steps_per_launch\": %.0f, "
               "\"gemm_ms_total\": %.4f, \"gemm_flops_total\": %.6e, \"gemm_calls\": %d, \"operands\": \"%s\", \"collection_during_learn\": %s",
               world, steps, warmup, envs, g_team, horizon, (long long)B, (long long)(B / 4), epochs, D,
               m.sec, m.agentSteps, m.agentSteps / m.sec, m.sec / steps * 1e3, m.consumeMs, fused ? "true" : "false", learner.UsesFreeRunningCollection() ? "free-running" : "lockstep",
               m.envMs, m.envLaunches, A, stepsPerLaunch, m.gemmMs, m.gemmFlops, m.gemmCalls, fp32 ? "fp32" : (std::getenv("RLGPU_AUTOCAST_FP16") ? "fp16 + dynamic loss scale" : "bf16"), overlap ? "true" : "false");
        // multi-GPU audit trail: how many RCCL ranks took part, every rank's own ms per iteration, and (rank 0) what one gradient all-reduce costs
        printf(", \"rccl_ranks\": %d, \"rank_ms_per_step\": [", world > 1 ? world : 0);
        for (size_t r = 0; r < m.rankSec.size(); r++) printf("%s%.4f", r ? ", " : "", m.rankSec[r] / steps * 1e3);
        printf("], \"allreduce_calls\": %d, \"allreduce_ms_per_optimizer_step\": %.5f", m.arCalls, m.arCalls ? m.arMs / m.arCalls : 0.0);
        // which exchange carried the gradients ("rccl" over xGMI; "shm" = the host-staged test transport, never a scaling result), and the switches seen
        printf(", \"transport\": \"%s\", \"env_overrides\": [%s]", world > 1 ? (transport_env && !strcmp(transport_env, "shm") ? "shm" : "rccl") : "none", overrides.c_str());
        if (g_userReward) printf(", \"user_reward\": \"a user RewardFunction subclass on %d host threads, %s\", \"host_threads\": %d, \"host_cores\": %u", cfg.numThreads,
                                 cfg.deferHostRewards ? "replayed after the fused launch (deferHostRewards)" : "between the device's steps", cfg.numThreads, std::thread::hardware_concurrency());
        if (!loadDir.empty()) printf(", \"policy\": \"loaded from a checkpoint\", \"mean_step_reward\": %.5f, \"policy_entropy\": %.4f", m.stepReward, m.entropy);
        if (haveTr)
            printf(", \"trained_regime\": {\"after_iterations\": %d, \"steps\": %d, \"value\": %.3f, \"ms_per_step\": %.6f, \"ppo_iter_ms\": %.6f, \"env_kernel_avg_ms\": %.4f}",
                   warmup + steps + trainedWarm, trainedSteps, tr.agentSteps / tr.sec, tr.sec / trainedSteps * 1e3, tr.consumeMs, tr.envLaunches ? tr.envMs / tr.envLaunches : 0.0);
        if (haveLn)
            printf(", \"trained_regime_learned\": {\"after_iterations\": %d, \"learning_epochs\": %d, \"steps\": %d, \"value\": %.3f, \"ms_per_step\": %.6f, \"ppo_iter_ms\": %.6f, \"env_kernel_avg_ms\": %.4f, "
                   "\"mean_step_reward\": %.5f, \"policy_entropy\": %.4f, \"mean_step_reward_fresh_policy\": %.5f, \"policy_entropy_fresh_policy\": %.4f}",
                   warmup + steps + trainedWarm + trainedSteps + learnedWarm, learnedEpochs, learnedSteps, ln.agentSteps / ln.sec, ln.sec / learnedSteps * 1e3, ln.consumeMs,
                   ln.envLaunches ? ln.envMs / ln.envLaunches : 0.0, ln.stepReward, ln.entropy, m.stepReward, m.entropy);
        printf("}\n");
        fflush(stdout);
    }
    return 0;
}
