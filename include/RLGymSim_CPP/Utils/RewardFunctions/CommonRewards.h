// The common rewards (SIM/Utils/RewardFunctions/CommonRewards.h:6-123) with the reference's constructor arguments and public
// fields.  Their arithmetic lives in the step kernel (rlgymppo_cpp_amd/csrc/arena_gym.h:compute_rewards).
#pragma once
#include "RewardFunction.h"
namespace RLGSC {
class EventReward : public RewardFunction {
public:
    struct WeightScales {
        float goal = 0, teamGoal = 0, concede = 0, assist = 0, touch = 0, shot = 0, shotPass = 0, save = 0, demo = 0, demoed = 0, boostPickup = 0;
        float& operator[](size_t i) { return (&goal)[i]; }
        float operator[](size_t i) const { return (&goal)[i]; }
    };
    WeightScales weights;
    EventReward(WeightScales scales) : weights(scales) {}
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override {
        for (int t = 0; t < cfg.n_terms; t++) if (cfg.terms[t].kind == RLGPU_RW_EVENT) return false;   // one event table per stack
        for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) cfg.event_weights[i] = weights[i];
        return PushTerm(cfg, RLGPU_RW_EVENT, weight, 0.f);
    }
};
class VelocityReward : public RewardFunction {
public:
    bool isNegative;
    VelocityReward(bool isNegative = false) : isNegative(isNegative) {}
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { return PushTerm(cfg, RLGPU_RW_VELOCITY, weight, isNegative ? 1.f : 0.f); }
};
class SaveBoostReward : public RewardFunction {
public:
    float exponent;
    SaveBoostReward(float exponent = 0.5f) : exponent(exponent) {}
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { return PushTerm(cfg, RLGPU_RW_SAVE_BOOST, weight, exponent); }
};
class VelocityBallToGoalReward : public RewardFunction {
public:
    bool ownGoal = false;
    VelocityBallToGoalReward(bool ownGoal = false) : ownGoal(ownGoal) {}
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { return PushTerm(cfg, RLGPU_RW_VEL_BALL_TO_GOAL, weight, ownGoal ? 1.f : 0.f); }
};
class VelocityPlayerToBallReward : public RewardFunction {
public:
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { return PushTerm(cfg, RLGPU_RW_VEL_PLAYER_TO_BALL, weight, 0.f); }
};
class FaceBallReward : public RewardFunction {
public:
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { return PushTerm(cfg, RLGPU_RW_FACE_BALL, weight, 0.f); }
};
class TouchBallReward : public RewardFunction {
public:
    float aerialWeight;
    TouchBallReward(float aerialWeight = 0) : aerialWeight(aerialWeight) {}
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { return PushTerm(cfg, RLGPU_RW_TOUCH_BALL, weight, aerialWeight); }
};
}
