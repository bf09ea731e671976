// StateSetter (SIM/Utils/StateSetters/StateSetter.h:5-10): ResetState writes the new episode's start into the arena and returns its
// GameState.  The built-in setters also have a device form (ApplyToDevice) that runs for all envs at once with the env's Philox stream;
// a user subclass has only ResetState, and the batched env then calls it on the host for every env whose episode ended, on an Arena
// facade (RocketSim/Arena.h) holding that env's downloaded state, and uploads the result.
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class StateSetter {
public:
    virtual GameState ResetState(Arena* arena) = 0;
    virtual bool ApplyToDevice(RlgpuGymConfig& cfg) const { return false; }
    virtual ~StateSetter() {}
};
}
