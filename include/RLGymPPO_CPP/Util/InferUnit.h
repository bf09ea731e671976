// InferUnit (PUB/Util/InferUnit.h:7-43, .cpp:11-132): one trained network loaded from a checkpoint file, fed by the user's
// OBSBuilder / ActionParser on host GameStates -- what deployment (the RLBot client) and evaluation code call.  The network
// runs on the GPU through include/rlgpu.h; `gpu = false` (libtorch's CPU path in the reference) is refused: this build has none.
#pragma once
#include "../Lists.h"
#include "../Threading/GameInst.h"
#include "../LearnerConfig.h"
namespace RLGPC {
class InferUnit {
public:
    RLGSC::OBSBuilder* obsBuilder;
    RLGSC::ActionParser* actionParser;
    bool isPolicy;

    InferUnit(RLGSC::OBSBuilder* obsBuilder, RLGSC::ActionParser* actionParser, std::filesystem::path modelPath, bool isPolicy, int obsSize,
              const IList& layerSizes, bool gpu = true);
    InferUnit(const InferUnit&) = delete;
    InferUnit& operator=(const InferUnit&) = delete;
    ~InferUnit();

    RLGSC::FList GetObs(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction);
    RLGSC::FList2 GetObs(const RLGSC::GameState& state, const RLGSC::ActionSet& prevActions);

    RLGSC::ActionSet InferPolicyAll(const RLGSC::GameState& state, const RLGSC::ActionSet& prevActions, bool deterministic, float temperature = 1.0f);
    RLGSC::Action InferPolicySingle(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction, bool deterministic,
                                    float temperature = 1.0f);
    RLGSC::FList InferPolicySingleDistrib(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction, float temperature = 1.0f);
    RLGSC::FList InferCriticAll(const RLGSC::GameState& state, const RLGSC::ActionSet& prevActions);
    float InferCriticSingle(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction);

private:
    struct Impl;
    Impl* impl;
};
}
