// Match (SIM/Envs/Match.h:13-55, Match.cpp:4-76): the bundle of plugins one env is built from, with the reference's member functions.
// Never owns its plugins (like the reference).
//
// Two ways a Match runs here.  (1) Its plugins describe themselves to the device (PlanDevice): the step kernel then does parsing, reward,
// obs, terminal check and reset for all envs at once.  (2) Any plugin KIND without a device form -- a user RewardFunction, OBSBuilder,
// TerminalCondition or StateSetter subclass -- is run on the host for every env by the Learner, through the functions below, exactly as
// Gym::Step calls them; the other kinds stay on the device (DevicePlan says which).
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

    // ---- the reference's host-side functions (Match.cpp) ----
    void EpisodeReset(const GameState& initialState) {
        prevActions = ActionSet(initialState.players.size());
        for (TerminalCondition* cond : terminalConditions) cond->Reset(initialState);
        rewardFn->Reset(initialState);
        obsBuilder->Reset(initialState);
    }
    FList2 BuildObservations(const GameState& state) {
        FList2 rows(state.players.size());
        obsBuilder->PreStep(state);
        for (size_t i = 0; i < state.players.size(); i++) rows[i] = obsBuilder->BuildOBS(state.players[i], state, prevActions[i]);
        return rows;
    }
    FList GetRewards(const GameState& state, bool done) {
        rewardFn->PreStep(state);
        return rewardFn->GetAllRewards(state, prevActions, done);
    }
    bool IsDone(const GameState& state) {   // short-circuit: a later condition is not even updated on a step an earlier one ended
        for (TerminalCondition* cond : terminalConditions) if (cond->IsTerminal(state)) return true;
        return false;
    }
    ScoreLine GetScoreLine(const GameState& state) { return state.scoreLine; }
    ActionSet ParseActions(const ActionParser::Input& actionsData, const GameState& gameState) {
        ActionSet actions = actionParser->ParseActions(actionsData, gameState);
        for (size_t i = 0; i < gameState.players.size() && i < actions.size(); i++) if (gameState.players[i].carState.isDemoed) actions[i] = {};   // demoed players get no input
        return actions;
    }
    GameState ResetState(Arena* arena) {
        GameState fresh = stateSetter->ResetState(arena);
        if ((int)fresh.players.size() != playerAmount)
            RG_ERR_CLOSE("Match::ResetState(): New state has a different amount of players, expected " << playerAmount << " but got " << fresh.players.size() << ".\n"
                         "Changing number of players at state reset is currently not supported.");
        for (BoostPad* pad : arena->_boostPads) pad->SetState({});
        return fresh;   // Match.cpp:55-69: the setter's GameState -- it predates the pad reset, so the new episode's first state shows the pads as the
                        // previous episode left them (unless the setter reset them itself, as KickoffState does through Arena::ResetToRandomKickoff)
    }

    // ---- device description ----
    struct DevicePlan {
        RlgpuGymConfig cfg;
        bool hostReward = false, hostTerminal = false, hostObs = false, hostSetter = false, hostParser = false;
        bool AnyHost() const { return hostReward || hostTerminal || hostObs || hostSetter || hostParser; }
    };
    // Which plugin kinds the step kernel runs and which stay on the host.  A kind on the host leaves a neutral device setting behind
    // (no reward terms, no conditions, no reset inside the step; a host parser's controls reach the kernel as rows, rlgpu_env_step_controls).
    DevicePlan PlanDevice(int tickSkip) const {
        DevicePlan plan; RlgpuGymConfig& cfg = plan.cfg;
        rlgpu_default_gym_config(&cfg);
        cfg.tick_skip = tickSkip;
        auto noReward = [&] { cfg.n_terms = 0; cfg.zero_sum = 0; for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) cfg.event_weights[i] = 0.f; };
        cfg.one_team = spawnOpponents ? 0 : 1;   // teamSize blue cars and nobody else: the env's orange slots stay empty
        if (!rewardFn || !obsBuilder || !actionParser || !stateSetter) RG_ERR_CLOSE("Match: a plugin is null");
        noReward();
        if (!rewardFn->AddDeviceTerms(cfg, 1.f)) { noReward(); plan.hostReward = true; }
        cfg.n_conds = 0;
        for (TerminalCondition* c : terminalConditions) if (!c->AddDeviceCondition(cfg)) plan.hostTerminal = true;
        if (plan.hostTerminal) cfg.n_conds = 0;
        if (!obsBuilder->ApplyToDevice(cfg)) plan.hostObs = true;
        if (!actionParser->ApplyToDevice(cfg)) { cfg.n_actions = actionParser->GetActionAmount(); plan.hostParser = true; }   // parsed on the host, stepped with rlgpu_env_step_controls
        if (!stateSetter->ApplyToDevice(cfg)) { cfg.setter_kind = RLGPU_SS_KICKOFF; plan.hostSetter = true; }
        // any plugin on the host: an env whose episode ended stays as it ended (the step kernel does not reset it) until the host has looked at it --
        // the new episode's first GameState shows the boost pads of that moment (Match.cpp:55-69) -- and has run the setter / asked for the reset
        if (plan.AnyHost()) cfg.host_resets = 1;
        return plan;
    }
    // the whole match as the device's gym configuration; throws when a plugin kind would have to run on the host
    RlgpuGymConfig ToDeviceConfig(int tickSkip) const {
        const DevicePlan plan = PlanDevice(tickSkip);
        if (plan.hostReward) RG_ERR_CLOSE("Match: the reward function has no device form (built-ins: CommonRewards.h, CombinedReward, ZeroSumReward; at most 8 terms, one EventReward)");
        if (plan.hostTerminal) RG_ERR_CLOSE("Match: a terminal condition has no device form (built-ins: NoTouchCondition, GoalScoreCondition)");
        if (plan.hostObs) RG_ERR_CLOSE("Match: the obs builder has no device form (built-ins: DefaultOBS, DefaultOBSPadded)");
        if (plan.hostSetter) RG_ERR_CLOSE("Match: the state setter has no device form (built-ins: RandomState, KickoffState)");
        if (plan.hostParser) RG_ERR_CLOSE("Match: the action parser has no device form (built-in: DiscreteAction)");
        return plan.cfg;
    }
};
}
