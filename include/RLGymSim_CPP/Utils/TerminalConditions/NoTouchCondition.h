// NoTouchCondition (SIM/Utils/TerminalConditions/NoTouchCondition.h:5-28): terminal after maxSteps steps without a ball touch
#pragma once
#include "TerminalCondition.h"
namespace RLGSC {
class NoTouchCondition : public TerminalCondition {
public:
    int stepsSinceTouch = 0, maxSteps;
    NoTouchCondition(int maxSteps) : maxSteps(maxSteps) {}
    bool AddDeviceCondition(RlgpuGymConfig& cfg) const override { cfg.no_touch_max_steps = maxSteps; return PushCond(cfg, RLGPU_TC_NO_TOUCH); }
};
typedef NoTouchCondition TimeoutCondition;
}
