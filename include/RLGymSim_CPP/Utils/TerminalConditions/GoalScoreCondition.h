// GoalScoreCondition (SIM/Utils/TerminalConditions/GoalScoreCondition.h:7-11)
#pragma once
#include "TerminalCondition.h"
namespace RLGSC {
class GoalScoreCondition : public TerminalCondition {
public:
    bool AddDeviceCondition(RlgpuGymConfig& cfg) const override { return PushCond(cfg, RLGPU_TC_GOAL_SCORE); }
};
}
