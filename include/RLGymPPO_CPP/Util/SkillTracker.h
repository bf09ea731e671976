// SkillTracker (PRIV/Util/SkillTracker.{h,cpp}): every `updateInterval` iterations the current policy plays `numEnvs` eval games
// against stored old versions of itself (deterministic actions, kickoff states, teams swapped at random per episode); every goal
// moves the ELO-style ratings (SkillTracker.cpp:72-86).  A new version is stored every `timestepsPerVersion` timesteps, at most
// `maxVersions` are kept.  The games are one small device env batch and the policies are device learner objects (include/rlgpu.h);
// only the few action indices and the goal flags of each step visit the host.
#pragma once
#include <map>
#include <random>
#include <set>
#include "SkillTrackerConfig.h"
#include "RenderSender.h"
namespace RLGPC {
struct SkillTracker {
    RenderSender* renderSender = NULL;
    struct Game {
        bool teamSwap = false;   // the old policy plays blue when set (SkillTracker.h:16-28)
        int oldPolicyIndex = 0;
    };
    std::vector<Game> games;
    SkillTrackerConfig config;
    struct RatingSet { std::map<std::string, float> data; };
    std::vector<RatingSet> oldRatings;            // one per stored version, same order
    int64_t timestepsSinceVersionMade = 0;
    uint64_t runCounter = 0;
    std::set<std::string> modeNames;
    RatingSet curRating;
    std::string modeName;                         // "<teamSize>v<teamSize>" of the eval env

    // `learner` = the training learner object whose policy is evaluated; layer description for the stored copies
    SkillTracker(const SkillTrackerConfig& config, rlgpu_learner* learner, int obsSize, int actionAmount, const IList& policyLayerSizes, int randomSeed,
                 RenderSender* renderSender = NULL);
    SkillTracker(const SkillTracker&) = delete;
    SkillTracker& operator=(const SkillTracker&) = delete;
    ~SkillTracker();

    void RunGames(int64_t timestepsDelta);                                     // SkillTracker.cpp:152-257
    void UpdateRatings(RatingSet& winner, RatingSet& loser, bool updateWinner, bool updateLoser, std::string mode);
    void AppendOldPolicy(const std::vector<float>& policyParams, RatingSet rating);   // e.g. from an old checkpoint (Learner.cpp:311-370)
    int NumOldPolicies() const;
    // "skill_rating" of RUNNING_STATS.json: an object per mode or one number (SkillTracker.cpp:259-291)
    RatingSet LoadRatingSet(const std::string& jsonValue, bool warn = true);
    std::string RatingsToJSON() const;

private:
    struct Impl; Impl* impl;
};
}
