// StateSetter (SIM/Utils/StateSetters/StateSetter.h:5-10).  The built-in setters run on the device with the env's Philox stream.
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class StateSetter {
public:
    virtual bool ApplyToDevice(RlgpuGymConfig& cfg) const { return false; }
    virtual ~StateSetter() {}
};
}
