// SkillTrackerConfig (PUB/Util/SkillTrackerConfig.h:7-44): fields kept so configs compile; the skill tracker itself is out of scope
#pragma once
namespace RLGPC {
struct SkillTrackerConfig {
    bool enabled = false;
    int numThreads = 16, numEnvsPerThread = 1;
    float simTime = 45, maxSimTime = 240;
    int updateInterval = 16;
    float ratingInc = 5, initialRating = 1000;
    int64_t timestepsPerVersion = 25 * 1000 * 1000;
    int maxVersions = 4;
    bool perModeRatings = true;
};
}
