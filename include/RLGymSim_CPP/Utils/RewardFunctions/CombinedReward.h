// CombinedReward (SIM/Utils/RewardFunctions/CombinedReward.h:6-53): weighted sum of child rewards
#pragma once
#include "RewardFunction.h"
namespace RLGSC {
class CombinedReward : public RewardFunction {
public:
    std::vector<RewardFunction*> rewardFuncs;
    std::vector<float> rewardWeights;
    bool ownsFuncs;
    CombinedReward(std::vector<RewardFunction*> funcs, std::vector<float> weights, bool ownsFuncs = false) : rewardFuncs(funcs), rewardWeights(weights), ownsFuncs(ownsFuncs) {
        if (funcs.size() != weights.size()) RG_ERR_CLOSE("CombinedReward: " << funcs.size() << " rewards but " << weights.size() << " weights");
    }
    CombinedReward(std::vector<std::pair<RewardFunction*, float>> funcsWithWeights, bool ownsFuncs = false) : ownsFuncs(ownsFuncs) {
        for (auto& fw : funcsWithWeights) { rewardFuncs.push_back(fw.first); rewardWeights.push_back(fw.second); }
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override {
        for (size_t i = 0; i < rewardFuncs.size(); i++) if (!rewardFuncs[i]->AddDeviceTerms(cfg, weight * rewardWeights[i])) return false;
        return true;
    }
    ~CombinedReward() override { if (ownsFuncs) for (auto f : rewardFuncs) delete f; }
};
}
