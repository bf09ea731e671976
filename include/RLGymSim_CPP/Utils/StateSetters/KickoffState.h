// KickoffState (SIM/Utils/StateSetters/KickoffState.h:7-10): Arena::ResetToRandomKickoff
#pragma once
#include "StateSetter.h"
namespace RLGSC {
class KickoffState : public StateSetter {
public:
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override { cfg.setter_kind = RLGPU_SS_KICKOFF; return true; }
};
}
