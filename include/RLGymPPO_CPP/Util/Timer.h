// Timer -- wall-clock stopwatch used for the per-iteration report ("Collection Time", "PPO Learn Time", ...).
// Same two calls as the reference's PUB/Util/Timer.h (Elapsed() in seconds as double, Reset()); that header only builds with MSVC
// (it mixes clock types), this one sticks to the monotonic clock.
#pragma once
#include "../Framework.h"

namespace RLGPC {

struct Timer {
    using Clock = std::chrono::steady_clock;
    Clock::time_point start;

    Timer() : start(Clock::now()) {}

    // seconds since construction or the last Reset()
    double Elapsed() const {
        const std::chrono::duration<double> since = Clock::now() - start;
        return since.count();
    }
    void Reset() { start = Clock::now(); }
};

}  // namespace RLGPC
