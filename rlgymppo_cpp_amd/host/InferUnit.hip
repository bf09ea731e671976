// InferUnit.hip -- RLGPC::InferUnit (PUB/Util/InferUnit.cpp:11-132) over the C-ABI: the checkpoint file is the reference's own
// TorchScript archive (rlgpu_lt_read_model), observations are built on the host by the user's OBSBuilder, the network runs on the
// device (rlgpu_policy_act / rlgpu_policy_probs / rlgpu_value_forward on a few rows).
#include <hip/hip_runtime.h>

#include <RLGymPPO_CPP/Util/InferUnit.h>

namespace RLGPC {

#define INFER_HIP(call)                                                                                  \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess) RG_ERR_CLOSE("InferUnit: " #call " failed: " << hipGetErrorString(_e));    \
    } while (0)

struct InferUnit::Impl {
    rlgpu_learner* net = nullptr;
    int obsSize = 0, actionAmount = 0, maxRows = 64;
    float *obs = nullptr, *outF = nullptr; int32_t* outI = nullptr;
    void Check(int rc, const char* what) { if (rc != RLGPU_OK) RG_ERR_CLOSE("InferUnit: rlgpu_" << what << " failed (" << rc << "): " << rlgpu_learner_last_error(net)); }
    // rows of observations -> device
    void Upload(const RLGSC::FList2& rows) {
        if ((int)rows.size() > maxRows) RG_ERR_CLOSE("InferUnit: " << rows.size() << " players in one state, at most " << maxRows);
        std::vector<float> flat;
        for (const RLGSC::FList& r : rows) {
            if ((int)r.size() != obsSize) RG_ERR_CLOSE("InferUnit: the OBS builder made " << r.size() << " values, the model was created for " << obsSize);
            flat.insert(flat.end(), r.begin(), r.end());
        }
        INFER_HIP(hipMemcpy(obs, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    }
};

InferUnit::InferUnit(RLGSC::OBSBuilder* obsBuilder_, RLGSC::ActionParser* actionParser_, std::filesystem::path modelPath, bool isPolicy_, int obsSize,
                     const IList& layerSizes, bool gpu)
    : obsBuilder(obsBuilder_), actionParser(actionParser_), isPolicy(isPolicy_), impl(new Impl()) {
    Impl& m = *impl;
    if (!gpu) RG_ERR_CLOSE("InferUnit: gpu = false: this build has no CPU inference path (the networks are HIP kernels)");
    if (!obsBuilder || !actionParser) RG_ERR_CLOSE("InferUnit: obsBuilder and actionParser must not be NULL");
    if (layerSizes.empty() || layerSizes.size() > 8) RG_ERR_CLOSE("InferUnit: 1..8 hidden layers");
    RG_LOG("InferUnit():");
    RG_LOG(" > Creating policy/critic...");
    m.obsSize = obsSize; m.actionAmount = actionParser->GetActionAmount();
    // one device learner object holds both networks; the one this unit is not about gets a minimal shape and is never run
    RlgpuLearnerConfig lc{};
    lc.obs_size = obsSize; lc.n_actions = m.actionAmount;
    lc.n_policy_layers = isPolicy ? (int)layerSizes.size() : 1; lc.n_critic_layers = isPolicy ? 1 : (int)layerSizes.size();
    for (int i = 0; i < 8; i++) { lc.policy_layers[i] = 16; lc.critic_layers[i] = 16; }
    for (size_t i = 0; i < layerSizes.size(); i++) (isPolicy ? lc.policy_layers : lc.critic_layers)[i] = layerSizes[i];
    lc.policy_lr = lc.critic_lr = 0; lc.ent_coef = 0; lc.clip_range = 0.2f; lc.temperature = 1; lc.use_bf16 = 0; lc.max_rows = m.maxRows;
    int rc = rlgpu_learner_create(&m.net, 0, &lc);
    m.Check(rc, "learner_create");
    RG_LOG(" > Loading policy/critic...");
    std::vector<int32_t> dims{obsSize};
    for (int h : layerSizes) dims.push_back(h);
    dims.push_back(isPolicy ? m.actionAmount : 1);
    std::vector<float> params((size_t)rlgpu_learner_num_params(m.net, isPolicy ? 0 : 1));
    if (rlgpu_lt_read_model(modelPath.string().c_str(), dims.data(), (int)dims.size() - 1, params.data()) != RLGPU_OK)
        RG_ERR_CLOSE("Failed to load model, checkpoint may be corrupt or of different model arch.\nException: " << rlgpu_lt_last_error());   // InferUnit.cpp:36-41
    m.Check(rlgpu_learner_set_params(m.net, isPolicy ? 0 : 1, params.data()), "learner_set_params");
    INFER_HIP(hipMalloc(&m.obs, (size_t)m.maxRows * obsSize * 4));
    INFER_HIP(hipMalloc(&m.outF, (size_t)m.maxRows * std::max(m.actionAmount, 1) * 4));
    INFER_HIP(hipMalloc(&m.outI, (size_t)m.maxRows * 4));
    RG_LOG(" > Done!");
}

InferUnit::~InferUnit() {
    if (impl->obs) (void)hipFree(impl->obs);
    if (impl->outF) (void)hipFree(impl->outF);
    if (impl->outI) (void)hipFree(impl->outI);
    if (impl->net) rlgpu_learner_destroy(impl->net);
    delete impl;
}

RLGSC::FList InferUnit::GetObs(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction) {
    return obsBuilder->BuildOBS(player, state, prevAction);
}
RLGSC::FList2 InferUnit::GetObs(const RLGSC::GameState& state, const RLGSC::ActionSet& prevActions) {
    if (prevActions.size() != state.players.size()) RG_ERR_CLOSE("InferUnit: " << prevActions.size() << " previous actions for " << state.players.size() << " players");
    RLGSC::FList2 rows;
    for (size_t i = 0; i < state.players.size(); i++) rows.push_back(obsBuilder->BuildOBS(state.players[i], state, prevActions[i]));
    return rows;
}

#define ASSERT_RIGHT_TYPE(ok, name, otherName) \
    if (!(ok)) RG_ERR_CLOSE("InferUnit: Failed to infer the " #name " because this inference unit was created to infer the " #otherName);

RLGSC::ActionSet InferUnit::InferPolicyAll(const RLGSC::GameState& state, const RLGSC::ActionSet& prevActions, bool deterministic, float temperature) {
    ASSERT_RIGHT_TYPE(isPolicy, policy, critic);
    Impl& m = *impl;
    RLGSC::FList2 rows = GetObs(state, prevActions);
    m.Upload(rows);
    m.Check(rlgpu_learner_set_temperature(m.net, temperature), "learner_set_temperature");
    m.Check(rlgpu_policy_act(m.net, m.obs, (int)rows.size(), deterministic ? 1 : 0, nullptr, m.outI, m.outF), "policy_act");
    m.Check(rlgpu_learner_sync(m.net), "learner_sync");
    RLGSC::IList picks(rows.size());
    INFER_HIP(hipMemcpy(picks.data(), m.outI, rows.size() * 4, hipMemcpyDeviceToHost));
    return actionParser->ParseActions(picks, state);
}

RLGSC::Action InferUnit::InferPolicySingle(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction, bool deterministic,
                                           float temperature) {
    ASSERT_RIGHT_TYPE(isPolicy, policy, critic);
    Impl& m = *impl;
    m.Upload({GetObs(player, state, prevAction)});
    size_t playerIndex = 0;   // the parser gets one index per player of the state; only this player's is meaningful (InferUnit.cpp:84-101)
    for (size_t i = 1; i < state.players.size(); i++)
        if (state.players[i].carId == player.carId) { playerIndex = i; break; }
    m.Check(rlgpu_learner_set_temperature(m.net, temperature), "learner_set_temperature");
    m.Check(rlgpu_policy_act(m.net, m.obs, 1, deterministic ? 1 : 0, nullptr, m.outI, m.outF), "policy_act");
    m.Check(rlgpu_learner_sync(m.net), "learner_sync");
    int32_t pick = 0;
    INFER_HIP(hipMemcpy(&pick, m.outI, 4, hipMemcpyDeviceToHost));
    RLGSC::IList picks(std::max<size_t>(state.players.size(), 1), 0);
    picks[playerIndex] = pick;
    return actionParser->ParseActions(picks, state)[playerIndex];
}

RLGSC::FList InferUnit::InferPolicySingleDistrib(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction, float temperature) {
    ASSERT_RIGHT_TYPE(isPolicy, policy, critic);
    Impl& m = *impl;
    m.Upload({GetObs(player, state, prevAction)});
    m.Check(rlgpu_learner_set_temperature(m.net, temperature), "learner_set_temperature");
    m.Check(rlgpu_policy_probs(m.net, m.obs, 1, m.outF), "policy_probs");
    m.Check(rlgpu_learner_sync(m.net), "learner_sync");
    RLGSC::FList probs(m.actionAmount);
    INFER_HIP(hipMemcpy(probs.data(), m.outF, probs.size() * 4, hipMemcpyDeviceToHost));
    return probs;
}

RLGSC::FList InferUnit::InferCriticAll(const RLGSC::GameState& state, const RLGSC::ActionSet& prevActions) {
    ASSERT_RIGHT_TYPE(!isPolicy, critic, policy);
    Impl& m = *impl;
    RLGSC::FList2 rows = GetObs(state, prevActions);
    m.Upload(rows);
    m.Check(rlgpu_value_forward(m.net, m.obs, (int)rows.size(), m.outF), "value_forward");
    m.Check(rlgpu_learner_sync(m.net), "learner_sync");
    RLGSC::FList values(rows.size());
    INFER_HIP(hipMemcpy(values.data(), m.outF, values.size() * 4, hipMemcpyDeviceToHost));
    return values;
}

float InferUnit::InferCriticSingle(const RLGSC::PlayerData& player, const RLGSC::GameState& state, const RLGSC::Action& prevAction) {
    ASSERT_RIGHT_TYPE(!isPolicy, critic, policy);
    Impl& m = *impl;
    m.Upload({GetObs(player, state, prevAction)});
    m.Check(rlgpu_value_forward(m.net, m.obs, 1, m.outF), "value_forward");
    m.Check(rlgpu_learner_sync(m.net), "learner_sync");
    float v = 0;
    INFER_HIP(hipMemcpy(&v, m.outF, 4, hipMemcpyDeviceToHost));
    return v;
}

}  // namespace RLGPC
