// The common rewards (SIM/Utils/RewardFunctions/CommonRewards.h:6-123) with the reference's constructor arguments and public
// fields.  Each has two forms: AddDeviceTerms describes it to the step kernel (rlgymppo_cpp_amd/csrc/arena_gym.h:compute_rewards), which is
// what training uses when the whole reward stack has a device form; GetReward is the host form, used when a user reward without a
// device form sits in the same stack (the Learner then evaluates the whole stack on the host) and by a standalone Gym.
#pragma once
#include <array>
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
    // host form (CommonRewards.cpp:9-47): weighted positive parts of the increase of eleven per-player values since the last call
    typedef std::array<float, RLGPU_NUM_EVENT_VALS> ValSet;
    std::unordered_map<uint32_t, ValSet> lastRegisteredValues;
    static ValSet ExtractValues(const PlayerData& p, const GameState& state) {
        const int own = (int)p.team;
        return ValSet{(float)p.matchGoals, (float)state.scoreLine[own], (float)state.scoreLine[1 - own], (float)p.matchAssists, (float)p.ballTouchedStep, (float)p.matchShots,
                      (float)p.matchShotPasses, (float)p.matchSaves, (float)p.matchDemos, (float)p.carState.isDemoed, p.boostFraction};
    }
    void Reset(const GameState& state) override {
        lastRegisteredValues.clear();
        for (const PlayerData& p : state.players) lastRegisteredValues[p.carId] = ExtractValues(p, state);
    }
    float GetReward(const PlayerData& player, const GameState& state, const Action&) override {
        ValSet& was = lastRegisteredValues[player.carId];
        const ValSet now = ExtractValues(player, state);
        float reward = 0;
        for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) reward += std::max(now[i] - was[i], 0.f) * weights[i];
        was = now;
        return reward;
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(EventReward)) return false;
        for (int t = 0; t < cfg.n_terms; t++) if (cfg.terms[t].kind == RLGPU_RW_EVENT) return false;   // one event table per stack
        for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) cfg.event_weights[i] = weights[i];
        return PushTerm(cfg, RLGPU_RW_EVENT, weight, 0.f);
    }
};
class VelocityReward : public RewardFunction {
public:
    bool isNegative;
    VelocityReward(bool isNegative = false) : isNegative(isNegative) {}
    float GetReward(const PlayerData& player, const GameState&, const Action&) override { return player.phys.vel.Length() / CommonValues::CAR_MAX_SPEED * (1 - 2 * (int)isNegative); }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(VelocityReward)) return false; return PushTerm(cfg, RLGPU_RW_VELOCITY, weight, isNegative ? 1.f : 0.f); }
};
class SaveBoostReward : public RewardFunction {
public:
    float exponent;
    SaveBoostReward(float exponent = 0.5f) : exponent(exponent) {}
    float GetReward(const PlayerData& player, const GameState&, const Action&) override { return std::min(std::max(std::pow(player.boostFraction, exponent), 0.f), 1.f); }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(SaveBoostReward)) return false; return PushTerm(cfg, RLGPU_RW_SAVE_BOOST, weight, exponent); }
};
class VelocityBallToGoalReward : public RewardFunction {
public:
    bool ownGoal = false;
    VelocityBallToGoalReward(bool ownGoal = false) : ownGoal(ownGoal) {}
    float GetReward(const PlayerData& player, const GameState& state, const Action&) override {
        const bool atOrange = (player.team == Team::BLUE) != ownGoal;
        const Vec toGoal = ((atOrange ? CommonValues::ORANGE_GOAL_BACK : CommonValues::BLUE_GOAL_BACK) - state.ball.pos).Normalized();
        return toGoal.Dot(state.ball.vel / CommonValues::BALL_MAX_SPEED);
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(VelocityBallToGoalReward)) return false; return PushTerm(cfg, RLGPU_RW_VEL_BALL_TO_GOAL, weight, ownGoal ? 1.f : 0.f); }
};
class VelocityPlayerToBallReward : public RewardFunction {
public:
    float GetReward(const PlayerData& player, const GameState& state, const Action&) override {
        return (state.ball.pos - player.phys.pos).Normalized().Dot(player.phys.vel / CommonValues::CAR_MAX_SPEED);
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(VelocityPlayerToBallReward)) return false; return PushTerm(cfg, RLGPU_RW_VEL_PLAYER_TO_BALL, weight, 0.f); }
};
class FaceBallReward : public RewardFunction {
public:
    float GetReward(const PlayerData& player, const GameState& state, const Action&) override {
        return player.carState.rotMat.forward.Dot((state.ball.pos - player.phys.pos).Normalized());
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(FaceBallReward)) return false; return PushTerm(cfg, RLGPU_RW_FACE_BALL, weight, 0.f); }
};
class TouchBallReward : public RewardFunction {
public:
    float aerialWeight;
    TouchBallReward(float aerialWeight = 0) : aerialWeight(aerialWeight) {}
    float GetReward(const PlayerData& player, const GameState& state, const Action&) override {
        return player.ballTouchedStep ? std::pow((state.ball.pos.z + CommonValues::BALL_RADIUS) / (CommonValues::BALL_RADIUS * 2), aerialWeight) : 0.f;
    }
    bool AddDeviceTerms(RlgpuGymConfig& cfg, float weight) const override { if (!RLG_IS_EXACTLY(TouchBallReward)) return false; return PushTerm(cfg, RLGPU_RW_TOUCH_BALL, weight, aerialWeight); }
};
}
