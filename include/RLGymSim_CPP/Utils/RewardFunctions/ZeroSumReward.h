// ZeroSumReward (SIM/Utils/RewardFunctions/ZeroSumReward.h:6-40, .cpp:3-29): team-spirit mixing minus the opponents' mean
#pragma once
#include "RewardFunction.h"
namespace RLGSC {
class ZeroSumReward : public RewardFunction {
public:
    RewardFunction* childFunc; float teamSpirit, opponentScale; bool ownsFunc;
    ZeroSumReward(RewardFunction* childFunc, float teamSpirit, float opponentScale = 1, bool ownsFunc = true)
        : childFunc(childFunc), teamSpirit(teamSpirit), opponentScale(opponentScale), ownsFunc(ownsFunc) {}
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override {
        if (cfg.n_terms != 0 || cfg.zero_sum) return false;   // the wrapper applies to the whole stack: it has to be the outermost reward
        if (!childFunc->AddDeviceTerms(cfg, weight)) return false;
        cfg.zero_sum = 1; cfg.team_spirit = teamSpirit; cfg.opp_scale = opponentScale;
        return true;
    }
    ~ZeroSumReward() override { if (ownsFunc) delete childFunc; }
};
}
