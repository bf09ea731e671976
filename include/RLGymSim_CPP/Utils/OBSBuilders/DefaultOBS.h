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
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override {
        cfg.pos_coef[0] = posCoef.x; cfg.pos_coef[1] = posCoef.y; cfg.pos_coef[2] = posCoef.z; cfg.vel_coef = velCoef; cfg.ang_vel_coef = angVelCoef;
        return true;
    }
};
}
