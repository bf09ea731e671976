// OBSBuilder (SIM/Utils/OBSBuilders/OBSBuilder.h:8-16)
#pragma once
#include "../Gamestates/GameState.h"
#include "../../../rlgpu.h"
namespace RLGSC {
class OBSBuilder {
public:
    virtual void Reset(const GameState& initialState) {}
    virtual void PreStep(const GameState& state) {}
    virtual FList BuildOBS(const PlayerData& player, const GameState& state, const Action& prevAction) { RG_ERR_CLOSE("OBSBuilder::BuildOBS() runs on the device for the built-in builders only"); }
    virtual bool ApplyToDevice(RlgpuGymConfig& cfg) const { return false; }
    virtual ~OBSBuilder() {}
};
}
