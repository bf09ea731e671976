// DefaultOBS (SIM/Utils/OBSBuilders/DefaultOBS.h:6-24, .cpp:3-55): ball(9) | prev action(8) | 34 pads | self(19) | mates | opponents
#pragma once
#include "OBSBuilder.h"
namespace RLGSC {
class DefaultOBS : public OBSBuilder {
public:
    Vec posCoef; float velCoef, angVelCoef;
    DefaultOBS(Vec posCoef = Vec(1 / CommonValues::SIDE_WALL_X, 1 / CommonValues::BACK_WALL_Y, 1 / CommonValues::CEILING_Z),
               float velCoef = 1 / CommonValues::CAR_MAX_SPEED, float angVelCoef = 1 / CommonValues::CAR_MAX_ANG_VEL)
        : posCoef(posCoef), velCoef(velCoef), angVelCoef(angVelCoef) {}
    // Host form (DefaultOBS.cpp:3-55), used off the hot path: InferUnit / deployment and the parity tests of the device builder.
    // One player block = pos*coef, forward, up, vel*coef, angVel*coef, boost fraction, on ground, has flip, demoed (19 floats).
    virtual void AddPlayerToOBS(FList& obs, const PlayerData& player, bool inv) {
        const PhysObj& phys = player.GetPhys(inv);
        obs += phys.pos * posCoef; obs += phys.rotMat.forward; obs += phys.rotMat.up; obs += phys.vel * velCoef; obs += phys.angVel * angVelCoef;
        obs += {player.boostFraction, (float)player.carState.isOnGround, (float)player.hasFlip, (float)player.carState.isDemoed};
    }
    // the part every DefaultOBS variant starts with: ball (9) | previous action (8) | pads (34); orange sees the mirrored field
    void AddSharedToOBS(FList& obs, const GameState& state, const Action& prevAction, bool inv) {
        const PhysObj& ball = state.GetBallPhys(inv);
        obs += ball.pos * posCoef; obs += ball.vel * velCoef; obs += ball.angVel * angVelCoef;
        for (int i = 0; i < Action::ELEM_AMOUNT; i++) obs += prevAction[i];
        const bool* pads = state.GetBoostPads(inv);
        for (int i = 0; i < CommonValues::BOOST_LOCATIONS_AMOUNT; i++) obs += (float)pads[i];
    }
    FList BuildOBS(const PlayerData& player, const GameState& state, const Action& prevAction) override {
        const bool inv = player.team == Team::ORANGE;
        FList obs, mates, opponents;
        AddSharedToOBS(obs, state, prevAction, inv);
        AddPlayerToOBS(obs, player, inv);
        for (const PlayerData& other : state.players)
            if (other.carId != player.carId) AddPlayerToOBS(other.team == player.team ? mates : opponents, other, inv);
        obs += mates; obs += opponents;
        return obs;
    }
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override { if (!RLG_IS_EXACTLY(DefaultOBS)) return false; WriteCoefs(cfg); return true; }
protected:
    void WriteCoefs(RlgpuGymConfig& cfg) const {
        cfg.pos_coef[0] = posCoef.x; cfg.pos_coef[1] = posCoef.y; cfg.pos_coef[2] = posCoef.z; cfg.vel_coef = velCoef; cfg.ang_vel_coef = angVelCoef;
    }
};
}
