// OBSBuilder -- what the networks see.
//
// Interface of the reference (SIM/Utils/OBSBuilders/OBSBuilder.h:8-16).  Two forms exist in this build:
//   * the DEVICE form: during training the observation rows are written by the step kernel straight into the experience tensors;
//     a builder describes itself through ApplyToDevice (coefficients, padding) and returns true -- the built-ins DefaultOBS and
//     DefaultOBSPadded do, the base class does not, and Match::ToDeviceConfig refuses a builder without one;
//   * the HOST form, BuildOBS on a GameState: used off the hot path (InferUnit / deployment, tests that pin the device rows
//     bit for bit).  The base implementation raises the framework's fatal error.
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"

namespace RLGSC {

class OBSBuilder {
public:
    virtual ~OBSBuilder() {}

    // episode / step hooks of the reference; the built-ins keep no state
    virtual void Reset(const GameState& initialState) { (void)initialState; }
    virtual void PreStep(const GameState& state) { (void)state; }

    virtual FList BuildOBS(const PlayerData& player, const GameState& state, const Action& prevAction) {
        (void)player; (void)state; (void)prevAction;
        RG_ERR_CLOSE("OBSBuilder::BuildOBS(): this builder has no host form");
    }

    virtual bool ApplyToDevice(RlgpuGymConfig& deviceCfg) const { (void)deviceCfg; return false; }
};

}  // namespace RLGSC
