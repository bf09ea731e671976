// TerminalCondition (SIM/Utils/TerminalConditions/TerminalCondition.h:6-9)
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class TerminalCondition {
public:
    virtual void Reset(const GameState& initialState) {}
    virtual bool IsTerminal(const GameState& currentState) { RG_ERR_CLOSE("TerminalCondition::IsTerminal() is evaluated on the device for the built-in conditions only"); }
    virtual bool AddDeviceCondition(RlgpuGymConfig& cfg) const { return false; }
    virtual ~TerminalCondition() {}
protected:
    static bool PushCond(RlgpuGymConfig& cfg, int kind) { if (cfg.n_conds >= 4) return false; cfg.conds[cfg.n_conds++] = kind; return true; }
};
}
