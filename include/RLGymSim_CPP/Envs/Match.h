// Match (SIM/Envs/Match.h:13-55): the bundle of plugins one env is built from.  Never owns its plugins (like the reference).
#pragma once
#include "../Utils/RewardFunctions/RewardFunction.h"
#include "../Utils/TerminalConditions/TerminalCondition.h"
#include "../Utils/OBSBuilders/OBSBuilder.h"
#include "../Utils/ActionParsers/ActionParser.h"
#include "../Utils/StateSetters/StateSetter.h"
namespace RLGSC {
class Match {
public:
    RewardFunction* rewardFn; std::vector<TerminalCondition*> terminalConditions; OBSBuilder* obsBuilder; ActionParser* actionParser; StateSetter* stateSetter;
    int teamSize; bool spawnOpponents; int playerAmount;
    ActionSet prevActions;
    Match(RewardFunction* rewardFn, std::vector<TerminalCondition*> terminalConditions, OBSBuilder* obsBuilder, ActionParser* actionParser,
          StateSetter* stateSetter, int teamSize = 1, bool spawnOpponents = true)
        : rewardFn(rewardFn), terminalConditions(terminalConditions), obsBuilder(obsBuilder), actionParser(actionParser), stateSetter(stateSetter),
          teamSize(teamSize), spawnOpponents(spawnOpponents), playerAmount(teamSize * (spawnOpponents ? 2 : 1)) { prevActions.resize(playerAmount); }
    // the whole match as the device's gym configuration; throws when a plugin has no device form
    RlgpuGymConfig ToDeviceConfig(int tickSkip) const {
        RlgpuGymConfig cfg; rlgpu_default_gym_config(&cfg);
        cfg.tick_skip = tickSkip; cfg.n_terms = 0; cfg.n_conds = 0; cfg.zero_sum = 0;
        for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) cfg.event_weights[i] = 0.f;
        if (!spawnOpponents) RG_ERR_CLOSE("Match: spawnOpponents = false is not supported by the batched env");
        if (!rewardFn || !rewardFn->AddDeviceTerms(cfg, 1.f)) RG_ERR_CLOSE("Match: the reward function has no device form (built-ins: CommonRewards.h, CombinedReward, ZeroSumReward; at most 8 terms, one EventReward)");
        for (auto c : terminalConditions) if (!c->AddDeviceCondition(cfg)) RG_ERR_CLOSE("Match: a terminal condition has no device form (built-ins: NoTouchCondition, GoalScoreCondition)");
        if (!obsBuilder || !obsBuilder->ApplyToDevice(cfg)) RG_ERR_CLOSE("Match: the obs builder has no device form (built-in: DefaultOBS)");
        if (!actionParser || !actionParser->ApplyToDevice(cfg)) RG_ERR_CLOSE("Match: the action parser has no device form (built-in: DiscreteAction)");
        if (!stateSetter || !stateSetter->ApplyToDevice(cfg)) RG_ERR_CLOSE("Match: the state setter has no device form (built-ins: RandomState, KickoffState)");
        return cfg;
    }
};
}
