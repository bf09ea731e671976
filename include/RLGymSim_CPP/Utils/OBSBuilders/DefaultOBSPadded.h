// DefaultOBSPadded (SIM/Utils/OBSBuilders/DefaultOBSPadded.h:6-24, .cpp:3-66): DefaultOBS with the teammate and opponent blocks
// padded to maxPlayers-1 / maxPlayers and shuffled per observation.  The device builds it for maxPlayers == team size (no zero
// blocks; same width as DefaultOBS); wider padding is refused by rlgpu_env_create.
#pragma once
#include "DefaultOBS.h"
namespace RLGSC {
class DefaultOBSPadded : public DefaultOBS {
public:
    int maxPlayers;
    DefaultOBSPadded(int maxPlayers, Vec posCoef = Vec(1 / CommonValues::SIDE_WALL_X, 1 / CommonValues::BACK_WALL_Y, 1 / CommonValues::CEILING_Z),
                     float velCoef = 1 / CommonValues::CAR_MAX_SPEED, float angVelCoef = 1 / CommonValues::CAR_MAX_ANG_VEL)
        : DefaultOBS(posCoef, velCoef, angVelCoef), maxPlayers(maxPlayers) {}
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override { DefaultOBS::ApplyToDevice(cfg); cfg.obs_max_players = maxPlayers; return maxPlayers > 0; }
};
}
