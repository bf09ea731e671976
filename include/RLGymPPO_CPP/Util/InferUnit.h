// InferUnit -- one trained network behind the user's observation builder and action parser: what a deployed bot or an evaluation
// harness calls once per decision.  API of the reference's PUB/Util/InferUnit.h:7-43 (constructor arguments, method names, argument
// order and return types), so code written against it compiles; behaviour follows PUB/Util/InferUnit.cpp:11-132.
//
//   * the network file is a PPO_POLICY.lt / PPO_CRITIC.lt archive (read without libtorch: rlgpu_lt_read_model);
//   * observations are built on the host by `obsBuilder`, actions decoded by `actionParser`;
//   * the forward pass runs on the GPU through include/rlgpu.h -- `gpu = false`, libtorch's CPU path in the reference, is refused
//     with an RG FATAL ERROR because this build has no CPU path.
#pragma once
#include "../Lists.h"
#include "../Threading/GameInst.h"
#include "../LearnerConfig.h"

namespace RLGPC {

class InferUnit {
    // spelled once, used below
    using State = RLGSC::GameState;
    using Player = RLGSC::PlayerData;
    using Act = RLGSC::Action;
    using Acts = RLGSC::ActionSet;
    using Row = RLGSC::FList;
    using Rows = RLGSC::FList2;

public:
    InferUnit(RLGSC::OBSBuilder* obsBuilder, RLGSC::ActionParser* actionParser, std::filesystem::path modelPath, bool isPolicy, int obsSize,
              const IList& layerSizes, bool gpu = true);
    ~InferUnit();
    InferUnit(const InferUnit&) = delete;
    InferUnit& operator=(const InferUnit&) = delete;

    // --- what the unit was built from (public in the reference, read by RLBotClient)
    RLGSC::OBSBuilder* obsBuilder;
    RLGSC::ActionParser* actionParser;
    bool isPolicy;   // false: the unit holds a critic (the reference keeps a NULL policy / critic pointer instead)

    // --- observations, exactly as the networks were trained on them
    Row GetObs(const Player& player, const State& state, const Act& prevAction);    // one player
    Rows GetObs(const State& state, const Acts& prevActions);                        // every player of the state, in order

    // --- policy units (calling these on a critic unit is the reference's "created to infer the critic" error)
    Acts InferPolicyAll(const State& state, const Acts& prevActions, bool deterministic, float temperature = 1.0f);
    Act InferPolicySingle(const Player& player, const State& state, const Act& prevAction, bool deterministic, float temperature = 1.0f);
    Row InferPolicySingleDistrib(const Player& player, const State& state, const Act& prevAction, float temperature = 1.0f);   // clamped probabilities

    // --- critic units
    Row InferCriticAll(const State& state, const Acts& prevActions);
    float InferCriticSingle(const Player& player, const State& state, const Act& prevAction);

private:
    struct Impl;   // device learner object + staging buffers (rlgymppo_cpp_amd/host/InferUnit.hip)
    Impl* impl;
};

}  // namespace RLGPC
