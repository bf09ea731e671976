// ZeroSumReward (SIM/Utils/RewardFunctions/ZeroSumReward.h:6-40, .cpp:3-29): team-spirit mixing minus the opponents' mean
#pragma once
#include "RewardFunction.h"
namespace RLGSC {
class ZeroSumReward : public RewardFunction {
public:
    RewardFunction* childFunc; float teamSpirit, opponentScale; bool ownsFunc;
    ZeroSumReward(RewardFunction* childFunc, float teamSpirit, float opponentScale = 1, bool ownsFunc = true)
        : childFunc(childFunc), teamSpirit(teamSpirit), opponentScale(opponentScale), ownsFunc(ownsFunc) {}
    // host form (ZeroSumReward.cpp:3-29)
    void Reset(const GameState& initialState) override { childFunc->Reset(initialState); }
    void PreStep(const GameState& state) override { childFunc->PreStep(state); }
    std::vector<float> GetAllRewards(const GameState& state, const ActionSet& prevActions, bool final) override {
        std::vector<float> r = childFunc->GetAllRewards(state, prevActions, final);
        int count[2] = {0, 0}; float mean[2] = {0.f, 0.f};
        for (size_t i = 0; i < state.players.size(); i++) { const int t = (int)state.players[i].team; count[t]++; mean[t] += r[i]; }
        for (int t = 0; t < 2; t++) mean[t] /= (float)std::max(count[t], 1);
        for (size_t i = 0; i < state.players.size(); i++) {
            const int t = (int)state.players[i].team;
            r[i] = r[i] * (1 - teamSpirit) + (mean[t] * teamSpirit) - (mean[1 - t] * opponentScale);
        }
        return r;
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(ZeroSumReward)) return false;
        if (cfg.n_terms != 0 || cfg.zero_sum) return false;   // the wrapper applies to the whole stack: it has to be the outermost reward
        if (!childFunc->AddDeviceTerms(cfg, weight)) return false;
        cfg.zero_sum = 1; cfg.team_spirit = teamSpirit; cfg.opp_scale = opponentScale;
        return true;
    }
    ~ZeroSumReward() override { if (ownsFunc) delete childFunc; }
};
}
