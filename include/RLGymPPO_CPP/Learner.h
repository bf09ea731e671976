// Learner (PUB/Learner.h, PUB/Learner.cpp:17-721): same construction, callbacks, Learn() loop, reports and checkpoint layout as the
// reference, driving ONE device env batch + the device learner of include/rlgpu.h instead of agent threads and libtorch.
#pragma once
#include "LearnerConfig.h"
#include "Threading/GameInst.h"
#include "Util/WelfordRunningStat.h"
#include "Util/Timer.h"
#include "Util/MetricSender.h"
#include "Util/RenderSender.h"
#include "Util/SkillTracker.h"
namespace RLGPC {
class Learner;
typedef std::function<void(Learner*, Report&)> IterationCallback;

// The reference's public header names three of its private classes as members (Learner.h:18-20).  Here they are the device objects of
// include/rlgpu.h that do those jobs: the PPO learner, the env batch that stands where the agent threads stood, the experience FIFO.
class PPOLearner { public: rlgpu_learner* device = nullptr; };
class ThreadAgentManager { public: rlgpu_env* device = nullptr; int numGames = 0; };
class ExperienceBuffer { public: rlgpu_expbuf* device = nullptr; };

class Learner {
public:
    LearnerConfig config;
    PPOLearner* ppo = nullptr;
    ThreadAgentManager* agentMgr = nullptr;
    ExperienceBuffer* expBuffer = nullptr;
    EnvCreateFn envCreateFn;
    int obsSize = 0, actionAmount = 0;
    std::string runID;
    uint64_t totalTimesteps = 0, totalEpochs = 0, totalIterations = 0;
    WelfordRunningStat returnStats;
    IterationCallback iterationCallback = nullptr;
    StepCallback stepCallback = nullptr;
    MetricSender* metricSender = nullptr;            // JSON-lines sender (Util/MetricSender.h); NULL unless config.sendMetrics
    SkillTracker* skillTracker = nullptr;            // ELO evaluation against stored old versions (Util/SkillTracker.h); NULL unless enabled
    RenderSender* renderSender = nullptr;            // RocketSimVis UDP sender (Util/RenderSender.h); NULL unless config.renderMode

    Learner(EnvCreateFn envCreateFn, LearnerConfig config);
    Learner(const Learner&) = delete;
    Learner& operator=(const Learner&) = delete;
    ~Learner();

    void Learn();                                    // Learner.cpp:436-606
    void UpdateLearningRates(float policyLR, float criticLR);
    std::vector<Report> GetAllGameMetrics();         // Learner.cpp:705-721 (and clears them, like the reference)
    void Save();                                     // <checkpointSaveFolder>/<totalTimesteps>/{PPO_*.lt, RUNNING_STATS.json}
    void Load();                                     // newest numbered subfolder of checkpointLoadFolder
    void SaveStats(std::filesystem::path path);
    void LoadStats(std::filesystem::path path);

    // one iteration, in the pieces Learn() strings together (also what tests drive)
    void CollectTimesteps();                         // ThreadAgentManager::CollectTimesteps for every game at once
    void AddNewExperience(Report& report);           // Learner.cpp:608-703: value predictions, GAE, return statistics
    void AddNewExperienceRagged(Report& report);     // ... of an iteration whose trajectories have their own lengths (free-running collection)
    void LearnPPO(Report& report);                   // PPOLearner::Learn (PPOLearner.cpp:67-349); with collectionDuringLearn it only LAUNCHES the epochs
    void FinishLearn(Report& report);                // ... and this waits for them and adds their statistics (Learn() calls it after the next collection)
    void LoadOldVersions(const std::vector<int32_t>& policyDims);   // Learner.cpp:311-370
    void RenderStep(int t);                          // ThreadAgent.cpp:164-186 for the first game
    int NumEnvs() const;
    int NumAgents() const;
    int StepsPerIteration() const;                   // T: steps every env takes per CollectTimesteps()
    uint32_t SamplerCalls() const;                   // launches the action sampler's counter-based stream has served (stored in RUNNING_STATS.json)
    uint32_t SamplerStream() const;                  // the action sampler's stream: the rank on a multi-GPU run (every rank explores on its own)
    // the experience of the last CollectTimesteps() copied to the host (tests, tools): obs [(T + 1) x agents x obsSize] -- row t + 1 is what
    // the policy sees after step t -- and actions / rewards / dones [T x agents]; agent row = env * players + slot.  Null = skip.
    void CopyCollected(std::vector<float>* obs, std::vector<int32_t>* actions, std::vector<float>* rewards, std::vector<int32_t>* dones);
    // free-running collection (LearnerConfig::lockstepCollection = false): the arrays above are laid out for StepCapacity() steps and game e's
    // players have CollectedSteps()[e] of them (rows beyond are stale); lockstep: every entry is StepsPerIteration()
    int StepCapacity() const;
    std::vector<int32_t> CollectedSteps() const;
    bool UsesFreeRunningCollection() const;
    uint64_t LastIterationTimesteps() const;         // agent-steps the last CollectTimesteps() gathered on this rank
    // multi-GPU (one process per GPU, launcher environment RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT; include/rlgpu.h rlgpu_comm_*):
    // envs are sharded over the ranks, ONE gradient all-reduce per optimizer step, rank 0 writes the checkpoints
    int Rank() const;
    int WorldSize() const;
    void CheckReplicas();                            // exchange errors and (every RLGPU_REPLICA_CHECK_EVERY = 50 iterations) a parameter checksum against rank 0's: RG_ERR_CLOSE on failure
    uint64_t ParamChecksum() const;                  // FNV-1a over the fp32 parameter bits of this rank
    // device-side clocks since the last reset: the env batch's step / collect launches and the learner's minibatch GEMM section (bench driver)
    void DeviceTimings(float& envMs, int& envLaunches, float& gemmMs, double& gemmFlops, int& gemmCalls, bool reset);
    bool UsesFusedCollection() const;
    double MaxOverRanks(double v);                   // collective; returns v on a single-GPU run
    std::vector<double> GatherOverRanks(double v);   // collective; every rank's v, indexed by rank
    // the gradient all-reduces timed with events on the learner's stream since DeviceTimings() switched the device clocks on
    void AllReduceTimings(float& ms, int& calls, bool reset);
private:
    struct Impl; Impl* impl;
};
}
