// RewardFunction (SIM/Utils/RewardFunctions/RewardFunction.h:7-38).  Same virtual interface as the reference.  Rewards run on
// the GPU inside the step kernel, so a reward must also be able to DESCRIBE itself as device reward terms (AddDeviceTerms);
// the built-ins of CommonRewards.h / CombinedReward.h / ZeroSumReward.h do.  A user subclass that only overrides GetReward
// cannot run in the batched env and Learner's constructor says so (there is no host fallback for the hot path).
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class RewardFunction {
public:
    virtual void Reset(const GameState& initialState) {}
    virtual void PreStep(const GameState& state) {}
    virtual float GetReward(const PlayerData& player, const GameState& state, const Action& prevAction) {
        RG_ERR_CLOSE("RewardFunction::GetReward() is not implemented by this reward");
    }
    virtual float GetFinalReward(const PlayerData& player, const GameState& state, const Action& prevAction) { return GetReward(player, state, prevAction); }
    virtual std::vector<float> GetAllRewards(const GameState& state, const ActionSet& prevActions, bool final) {
        std::vector<float> r(state.players.size());
        for (size_t i = 0; i < r.size(); i++) r[i] = final ? GetFinalReward(state.players[i], state, prevActions[i]) : GetReward(state.players[i], state, prevActions[i]);
        return r;
    }
    // append this reward, scaled by `weight`, to the device reward stack; false = not expressible on the device
    virtual bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const { return false; }
    virtual ~RewardFunction() {}
protected:
    static bool PushTerm(RlgpuGymConfig& cfg, int kind, float weight, float p0) {
        if (cfg.n_terms >= 8) return false;
        cfg.terms[cfg.n_terms].kind = kind; cfg.terms[cfg.n_terms].weight = weight; cfg.terms[cfg.n_terms].p0 = p0; cfg.n_terms++;
        return true;
    }
};
}
