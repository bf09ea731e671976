// NoTouchCondition (SIM/Utils/TerminalConditions/NoTouchCondition.h:5-28): terminal after maxSteps steps without a ball touch
#pragma once
#include "TerminalCondition.h"
namespace RLGSC {
class NoTouchCondition : public TerminalCondition {
public:
    int stepsSinceTouch = 0, maxSteps;
    NoTouchCondition(int maxSteps) : maxSteps(maxSteps) {}
    void Reset(const GameState&) override { stepsSinceTouch = 0; }
    bool IsTerminal(const GameState& currentState) override {   // host form: the counter restarts on any player's touch
        for (const PlayerData& player : currentState.players)
            if (player.ballTouchedStep) { stepsSinceTouch = 0; return false; }
        return ++stepsSinceTouch >= maxSteps;
    }
    bool AddDeviceCondition(RlgpuGymConfig& cfg) const override { if (!RLG_IS_EXACTLY(NoTouchCondition)) return false; cfg.no_touch_max_steps = maxSteps; return PushCond(cfg, RLGPU_TC_NO_TOUCH); }
};
typedef NoTouchCondition TimeoutCondition;
}
