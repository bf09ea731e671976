// GoalScoreCondition -- the episode ends on the step in which a goal is scored.
//
// The reference compares the score line of the new GameState with the one it remembered
// (SIM/Utils/TerminalConditions/GoalScoreCondition.h:7-11).  The device does the same inside the step kernel (compute_done in
// csrc/arena_gym.h, condition kind RLGPU_TC_GOAL_SCORE); this class only appends that kind to the env's condition list, in the order
// the conditions were given to Match -- the order matters because the reference short-circuits: a later condition is not even
// updated on a step an earlier one ended.
#pragma once
#include "TerminalCondition.h"

namespace RLGSC {

class GoalScoreCondition : public TerminalCondition {
public:
    bool IsTerminal(const GameState& currentState) override { return Math::IsBallScored(currentState.ball.pos); }   // host form

    bool AddDeviceCondition(RlgpuGymConfig& deviceCfg) const override { if (!RLG_IS_EXACTLY(GoalScoreCondition)) return false;
        return PushCond(deviceCfg, RLGPU_TC_GOAL_SCORE);
    }
};

}  // namespace RLGSC
