// SkillTrackerConfig: every field name and default of PUB/Util/SkillTrackerConfig.h:7-49
#pragma once
#include "../Lists.h"
#include "../Threading/GameInst.h"
namespace RLGPC {
struct SkillTrackerConfig {
    bool enabled = false;
    EnvCreateFn envCreateFunc = NULL;        // env description for the eval games; NULL -> the learner's
    StepCallback stepCallback = NULL;        // NULL -> the learner's
    int numEnvs = 4;                         // eval games (one small device batch)
    float simTime = 60;                      // simulated seconds per evaluation, shared by the games
    int updateInterval = 4;                  // iterations between evaluations
    int64_t timestepsPerVersion = 50 * 1000 * 1000;
    int maxVersions = 4;
    int numThreads = 8;                      // unused: the eval games step together as one batch
    bool perModeRatings = true;
    bool loadOldVersionsFromCheckpoints = true;
    bool startWithVersion = true;
    bool kickoffStatesOnly = true;
    float ratingInc = 5;
    float initialRating = 1000;
};
}
