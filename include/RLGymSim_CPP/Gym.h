// Gym (SIM/Gym.h:5-35, Gym.cpp:40-102).  The reference's one-arena environment, whole: constructor arguments, public fields, Reset()
// and Step() with the reference's return values (Step's obs is the TERMINAL observation when the episode ends; the caller resets).
//
// The Learner does not step Gyms: it reads match and tickSkip from the one EnvCreateFn returns and runs all games as one device batch.
// A Gym used directly -- evaluation code, tests, tools written against RLGymSim_CPP -- steps its Arena facade through a one-env device
// batch it creates on first use (rlgymppo_cpp_amd/host/Gym.hip): physics, event tracker and match counters run in the step kernel, the
// match's plugins run here through their host forms, in Gym::Step's order.
#pragma once
#include "Envs/Match.h"
namespace RLGSC {
// GameEventTracker (RS/Sim/GameEventTracker/GameEventTracker.h): its state lives in the env's gym block on the device (shot / goal / save
// detection runs in the step kernel); the member exists for source compatibility
struct GameEventTracker { void ResetPersistentInfo() {} };

class Gym {
public:
    Arena* arena;
    GameEventTracker eventTracker;
    Match* match; int tickSkip; int actionDelay;
    GameState prevState;
    std::vector<uint32_t> carIds;
    int totalTicks = 0, totalSteps = 0;
    struct StepResult { FList2 obs; FList reward; bool done = false; GameState state; };

    Gym(Match* match, int tickSkip, CarConfig carConfig = CAR_CONFIG_OCTANE, GameMode gameMode = GameMode::SOCCAR, MutatorConfig mutatorConfig = MutatorConfig(GameMode::SOCCAR));
    Gym(const Gym&) = delete;
    Gym& operator=(const Gym&) = delete;
    virtual FList2 Reset();
    virtual StepResult Step(const ActionParser::Input& actionsData);
    virtual ~Gym();
private:
    struct Device; Device* dev = nullptr;
};
}
