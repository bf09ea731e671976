// DiscreteAction (SIM/Utils/ActionParsers/DiscreteAction.h:14-23, .cpp:3-67): the 90-row lookup table (24 ground + 66 aerial)
#pragma once
#include "ActionParser.h"
namespace RLGSC {
class DiscreteAction : public ActionParser {
public:
    std::vector<Action> actions;
    DiscreteAction() {
        float rows[128 * 8];
        int n = rlgpu_action_table(rows, 128);
        actions.resize(n);
        for (int i = 0; i < n; i++) for (int k = 0; k < 8; k++) actions[i][k] = rows[i * 8 + k];
    }
    ActionSet ParseActions(const Input& actionsData, const GameState& state) override {
        ActionSet out;
        for (int idx : actionsData) {
            if (idx < 0 || idx >= (int)actions.size()) RG_ERR_CLOSE("DiscreteAction: action index " << idx << " out of range");
            out.push_back(actions[idx]);
        }
        return out;
    }
    int GetActionAmount() override { return (int)actions.size(); }
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override { if (!RLG_IS_EXACTLY(DiscreteAction)) return false; cfg.n_actions = (int)actions.size(); return true; }
};
}
