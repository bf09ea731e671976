// TerminalCondition (SIM/Utils/TerminalConditions/TerminalCondition.h:6-9).  Same interface as the reference; AddDeviceCondition is how
// a condition tells the batched env that the step kernel can evaluate it.  A user subclass that only overrides IsTerminal runs on the
// host: the Learner then evaluates ALL of the match's conditions there, per env and step, in Match::IsDone's order.
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class TerminalCondition {
public:
    virtual void Reset(const GameState& initialState) {}
    virtual bool IsTerminal(const GameState& currentState) = 0;
    virtual bool AddDeviceCondition(RlgpuGymConfig& cfg) const { return false; }
    virtual ~TerminalCondition() {}
protected:
    static bool PushCond(RlgpuGymConfig& cfg, int kind) { if (cfg.n_conds >= 4) return false; cfg.conds[cfg.n_conds++] = kind; return true; }
};
}
