// ActionParser (SIM/Utils/ActionParsers/ActionParser.h:9-15)
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class ActionParser {
public:
    typedef IList Input;
    virtual ActionSet ParseActions(const Input& actionsData, const GameState& state) = 0;
    virtual int GetActionAmount() = 0;
    virtual bool ApplyToDevice(RlgpuGymConfig& cfg) const { return false; }
    virtual ~ActionParser() {}
};
}
