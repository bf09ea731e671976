// KickoffState (SIM/Utils/StateSetters/KickoffState.h:7-10): Arena::ResetToRandomKickoff
#pragma once
#include "StateSetter.h"
namespace RLGSC {
class KickoffState : public StateSetter {
public:
    GameState ResetState(Arena* arena) override { arena->ResetToRandomKickoff(); return GameState(arena); }
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override { if (!RLG_IS_EXACTLY(KickoffState)) return false; cfg.setter_kind = RLGPU_SS_KICKOFF; return true; }
};
}
