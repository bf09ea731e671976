// CombinedReward -- a weighted sum of child rewards (SIM/Utils/RewardFunctions/CombinedReward.h:6-53).
//
// Both constructor forms of the reference exist: two parallel vectors, or the { {reward, weight}, ... } list the example program
// uses.  Like the reference the children are NOT owned unless `ownsFuncs` is set.  On the device a CombinedReward is not an object
// but up to 8 (kind, weight, parameter) terms of the env's gym configuration: AddDeviceTerms() asks every child to append its
// term(s) with the child's weight multiplied in, so nested CombinedRewards flatten.
#pragma once
#include "RewardFunction.h"

namespace RLGSC {

class CombinedReward : public RewardFunction {
public:
    std::vector<RewardFunction*> rewardFuncs;
    std::vector<float> rewardWeights;
    bool ownsFuncs;

    CombinedReward(std::vector<RewardFunction*> funcs, std::vector<float> weights, bool owns = false)
        : rewardFuncs(std::move(funcs)), rewardWeights(std::move(weights)), ownsFuncs(owns) {
        if (rewardFuncs.size() != rewardWeights.size())
            RG_ERR_CLOSE("CombinedReward: " << rewardFuncs.size() << " rewards but " << rewardWeights.size() << " weights");
    }

    CombinedReward(std::vector<std::pair<RewardFunction*, float>> weighted, bool owns = false) : ownsFuncs(owns) {
        rewardFuncs.reserve(weighted.size());
        rewardWeights.reserve(weighted.size());
        for (const auto& entry : weighted) {
            rewardFuncs.push_back(entry.first);
            rewardWeights.push_back(entry.second);
        }
    }

    ~CombinedReward() override {
        if (!ownsFuncs) return;
        for (RewardFunction* child : rewardFuncs) delete child;
    }

    // host form (CombinedReward.h:27-45): every child sees every hook; the children's reward vectors add up, weighted
    void Reset(const GameState& initialState) override { for (RewardFunction* child : rewardFuncs) child->Reset(initialState); }
    void PreStep(const GameState& state) override { for (RewardFunction* child : rewardFuncs) child->PreStep(state); }
    std::vector<float> GetAllRewards(const GameState& state, const ActionSet& prevActions, bool final) override {
        std::vector<float> total(state.players.size(), 0.f);
        for (size_t k = 0; k < rewardFuncs.size(); k++) {
            const std::vector<float> part = rewardFuncs[k]->GetAllRewards(state, prevActions, final);
            for (size_t j = 0; j < part.size() && j < total.size(); j++) total[j] += part[j] * rewardWeights[k];
        }
        return total;
    }

    bool AddDeviceTerms(RlgpuGymConfig& deviceCfg, float outerWeight) const override { if (!RLG_IS_EXACTLY(CombinedReward)) return false;
        for (size_t k = 0; k < rewardFuncs.size(); k++) {
            const bool ok = rewardFuncs[k]->AddDeviceTerms(deviceCfg, outerWeight * rewardWeights[k]);
            if (!ok) return false;   // a child without a device form: Match::ToDeviceConfig turns this into the fatal error
        }
        return true;
    }
};

}  // namespace RLGSC
