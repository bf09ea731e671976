// DefaultOBSPadded (SIM/Utils/OBSBuilders/DefaultOBSPadded.h:6-24, .cpp:3-66): DefaultOBS with the teammate and opponent blocks
// padded to maxPlayers-1 / maxPlayers and shuffled per observation.  The device builds it for team size <= maxPlayers <= 4
// (rlgpu_env_create refuses anything else, like the reference's "Too many teammates" error).
#pragma once
#include <algorithm>
#include <random>
#include "DefaultOBS.h"
namespace RLGSC {
class DefaultOBSPadded : public DefaultOBS {
public:
    int maxPlayers;
    DefaultOBSPadded(int maxPlayers, Vec posCoef = Vec(1 / CommonValues::SIDE_WALL_X, 1 / CommonValues::BACK_WALL_Y, 1 / CommonValues::CEILING_Z),
                     float velCoef = 1 / CommonValues::CAR_MAX_SPEED, float angVelCoef = 1 / CommonValues::CAR_MAX_ANG_VEL)
        : DefaultOBS(posCoef, velCoef, angVelCoef), maxPlayers(maxPlayers) {}
    // Host form (DefaultOBSPadded.cpp:3-66): zero blocks up to maxPlayers-1 mates / maxPlayers opponents, both lists shuffled per call.
    // The reference shuffles with RocketSim's process-wide engine; this builder owns one.
    std::default_random_engine shuffleEngine{0};
    FList BuildOBS(const PlayerData& player, const GameState& state, const Action& prevAction) override {
        const bool inv = player.team == Team::ORANGE;
        FList obs;
        AddSharedToOBS(obs, state, prevAction, inv);
        const size_t before = obs.size();
        AddPlayerToOBS(obs, player, inv);
        const size_t blockSize = obs.size() - before;
        FList2 mates, opponents;
        for (const PlayerData& other : state.players) {
            if (other.carId == player.carId) continue;
            FList block;
            AddPlayerToOBS(block, other, inv);
            (other.team == player.team ? mates : opponents).push_back(block);
        }
        if ((int)mates.size() > maxPlayers - 1) RG_ERR_CLOSE("DefaultOBSPadded: Too many teammates for OBS, maximum is " << (maxPlayers - 1));
        if ((int)opponents.size() > maxPlayers) RG_ERR_CLOSE("DefaultOBSPadded: Too many opponents for OBS, maximum is " << maxPlayers);
        opponents.resize(maxPlayers, FList(blockSize)); mates.resize(maxPlayers - 1, FList(blockSize));
        std::shuffle(mates.begin(), mates.end(), shuffleEngine);
        std::shuffle(opponents.begin(), opponents.end(), shuffleEngine);
        for (const FList& b : mates) obs += b;
        for (const FList& b : opponents) obs += b;
        return obs;
    }
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override { if (!RLG_IS_EXACTLY(DefaultOBSPadded)) return false; WriteCoefs(cfg); cfg.obs_max_players = maxPlayers; return maxPlayers > 0; }
};
}
