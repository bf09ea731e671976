// GameInst (PUB/Threading/GameInst.h:7-57): the per-game handle callbacks receive.  The games themselves are rows of one device
// batch; a GameInst carries that game's index, its metrics Report and the reward trackers the reference keeps per game.
#pragma once
#include <RLGymSim_CPP/Gym.h>
#include "../Util/Report.h"
#include "../Util/AvgTracker.h"
namespace RLGPC {
struct GameInst;
typedef std::function<void(GameInst*, const RLGSC::Gym::StepResult&, Report&)> StepCallback;
struct EnvCreateResult { RLGSC::Match* match; RLGSC::Gym* gym; };   // GameInst.h:9-14
typedef std::function<EnvCreateResult()> EnvCreateFn;
struct GameInst {
    RLGSC::Gym* gym = nullptr; RLGSC::Match* match = nullptr;   // shared descriptors (owned by the Learner)
    int index = 0;                                              // env index inside the device batch
    uint64_t totalSteps = 0;
    float curEpRew = 0; AvgTracker avgStepRew, avgEpRew;
    Report _metrics;
    StepCallback stepCallback = nullptr;
    bool isEval = false;
    const Report& GetMetrics() const { return _metrics; }
    void ResetMetrics() { _metrics.Clear(); }
};
}
