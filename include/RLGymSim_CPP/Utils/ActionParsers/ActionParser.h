// ActionParser -- turns the integers the policy emits into car controls.
//
// Interface of the reference (SIM/Utils/ActionParsers/ActionParser.h:9-15): `Input` is the list of per-player policy outputs,
// ParseActions maps it to one Action (8 floats) per player, GetActionAmount is the size of the policy's output layer.  In this
// build the parsing of a training step happens inside the step kernel (a lookup in the device copy of the action table), so a
// parser also has to describe itself to the device: ApplyToDevice fills the table size into the gym configuration and returns
// true; the base class returns false, which makes Match::ToDeviceConfig refuse a parser that has no device form.  The host form
// (ParseActions) is still what InferUnit, the skill tracker's renderer and tests call.
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"

namespace RLGSC {

class ActionParser {
public:
    typedef IList Input;

    virtual ~ActionParser() {}

    virtual int GetActionAmount() = 0;
    virtual ActionSet ParseActions(const Input& actionsData, const GameState& state) = 0;

    // device description; false = "I only exist on the host"
    virtual bool ApplyToDevice(RlgpuGymConfig& deviceCfg) const { (void)deviceCfg; return false; }
};

}  // namespace RLGSC
