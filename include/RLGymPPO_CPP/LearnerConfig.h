// LearnerConfig: every field name and default of PUB/LearnerConfig.h:14-80
#pragma once
#include "Lists.h"
#include "PPO/PPOLearnerConfig.h"
#include "Util/SkillTrackerConfig.h"
namespace RLGPC {
enum class LearnerDeviceType { AUTO, CPU, GPU_CUDA };
struct LearnerConfig {
    int numThreads = 8;                 // envs = numThreads * numGamesPerThread (one device batch; there are no host threads)
    int numGamesPerThread = 16;
    int minInferenceSize = 80;          // unused: inference always covers the whole batch
    bool renderMode = false;            // one game paced to real time, states to RocketSimVis over UDP (Util/RenderSender.h)
    float renderTimeScale = 1.5f;
    bool renderDuringTraining = false;
    uint64_t timestepLimit = 0;
    int64_t expBufferSize = 100 * 1000;
    int64_t timestepsPerIteration = 50 * 1000;
    bool standardizeReturns = true;
    bool standardizeOBS = false;
    int maxReturnsPerStatsInc = 150;
    int stepsPerObsStatsInc = 5;
    bool deterministic = false;
    bool collectionDuringLearn = false; // the PPO epochs of iteration k run on their own stream while iteration k + 1 is collected (its report carries their statistics)
    PPOLearnerConfig ppo = {};
    float gaeLambda = 0.95f;
    float gaeGamma = 0.99f;
    float rewardClipRange = 10;
    std::filesystem::path checkpointLoadFolder = "checkpoints";
    std::filesystem::path checkpointSaveFolder = "checkpoints";
    bool saveFolderAddUnixTimestamp = false;
    int64_t timestepsPerSave = 500 * 1000;
    int randomSeed = 123;
    int checkpointsToKeep = 5;
    LearnerDeviceType deviceType = LearnerDeviceType::AUTO;   // AUTO / GPU_CUDA = the HIP device; CPU is refused (no CPU path)
    bool sendMetrics = true;            // JSON lines under metrics/<project>/<run id>.jsonl (Util/MetricSender.h); tools/metric_receiver.py -> wandb
    std::string metricsProjectName = "rlgymppo-cpp";
    std::string metricsGroupName = "unnamed-runs";
    std::string metricsRunName = "rlgymppo-cpp-run";
    SkillTrackerConfig skillTrackerConfig = {};
    // ---- appended by this build (after every reference field, so aggregate initialisers written for the reference keep their meaning) ----
    // 0 = the reference's ComputeGAE as is: at a truncated trajectory end the bootstrap value is the NEXT ROW of the concatenated batch,
    // i.e. the first state of the neighbouring trajectory (TorchFuncs.cpp:36); 1 = the agent's own V(s_T)
    int gaeNextValueMode = 0;
    // The example program's per-step metrics without a step callback: the step kernels accumulate them from every step's GameState and each
    // iteration's report gets "player_speed" (mean |car velocity|, uu/s), "ball_touch_ratio" and "in_air_ratio" (fractions of player-steps)
    // -- collection stays in one launch instead of leaving the device every step (rlgpu_env_enable_step_stats)
    bool deviceStepMetrics = false;
    // false (default) = collection as the reference's agent threads run it (ThreadAgentManager.cpp:16-82, ThreadAgent.cpp:57-59): every wavefront
    // of the collection launch steps its games at its own pace until the batch has timestepsPerIteration steps TOGETHER, and an iteration's
    // trajectories have whatever length each game reached (a launch does not wait for its slowest game; an iteration holds between
    // timestepsPerIteration and timestepsPerIteration + one step of every game).  Used when the whole batch is resident on the device at once and
    // no plugin or callback needs the host per step; otherwise -- and with true -- every game makes the same number of steps per iteration
    // (reproducible from the seed alone: the free-running iteration's trajectory LENGTHS depend on timing, their contents do not).
    bool lockstepCollection = false;
    // (added) The reference sums its gradients in a fixed order (single stream, PPOLearner.cpp:205-215); so do this build's fused minibatch kernels
    // (per-slab partials added in slab order).  What still varies from run to run by default is how many steps each game contributes to a
    // free-running iteration, and -- in fp32 mode or with nets wider than 256 -- the order of the per-layer kernels' atomics.  true: lockstep
    // collection AND fixed-order sums everywhere (rlgpu_learner_set_deterministic): the run is a function of its seed, bit for bit, on one rank
    // or several.
    bool deterministicGradients = false;
    // lockstep collection of a batch with more wavefront-groups than the GPU keeps resident: -1 = through the step queue (rlgpu_env_set_collect_queue;
    // same results, the launch does not wait for the slot that got the slow groups), 0 = one workgroup per group as before, 1 = the queue always
    int collectStepQueue = -1;
    // (added, round 6) A user RewardFunction and / or a step callback as the ONLY host work: true (default) = collection stays in one fused launch, the kernel
    // stores every step's GameState source (rlgpu_env_enable_step_records: 336 B per 1v1 env and step), and after the launch every env's steps are replayed in
    // order on numThreads host threads -- Match::GetRewards (PreStep, GetAllRewards(state, prevActions, final)), the plugins' Reset at episode starts, GameInst's
    // bookkeeping, the callback -- before the value pass reads the rewards.  A reward does not feed the next action, so the experience is the per-step host
    // path's (tests/cpp/plugin_fallback_check.cpp).  false = such plugins run between the device's steps, like every other host kind (obs builder, terminal
    // conditions, state setter, action parser always do).
    bool deferHostRewards = true;
};
}
