// Gym (SIM/Gym.h:5-35).  In the reference one Gym owns one Arena; here a Gym is the description (match + tickSkip) the
// batched device env is created from, plus the StepResult type callbacks receive.
#pragma once
#include "Envs/Match.h"
namespace RLGSC {
class Gym {
public:
    Match* match; int tickSkip; int actionDelay; uint64_t totalTicks = 0, totalSteps = 0;
    GameState prevState;
    struct StepResult { FList2 obs; FList reward; bool done = false; GameState state; };
    Gym(Match* match, int tickSkip) : match(match), tickSkip(tickSkip), actionDelay(tickSkip - 1) {}
    virtual ~Gym() {}
};
}
