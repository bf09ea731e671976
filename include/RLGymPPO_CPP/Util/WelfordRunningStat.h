// WelfordRunningStat (PUB/Util/WelfordRunningStat.h:5-84) for the scalar return statistic: per-sample Welford in double
#pragma once
#include "../Framework.h"
namespace RLGPC {
struct WelfordRunningStat {
    double runningMean = 0, runningVariance = 0; int64_t count = 0;
    void Increment(const FList& samples, int num) {
        for (int i = 0; i < num && i < (int)samples.size(); i++) {
            double delta = samples[i] - runningMean, deltaN = delta / (double)(count + 1);
            runningMean += deltaN;
            runningVariance += delta * deltaN * (double)count;
            count++;
        }
    }
    double GetMean() const { return count < 2 ? 0.0 : runningMean; }
    double GetSTD() const {
        if (count < 2) return 1.0;
        double var = runningVariance / (double)(count - 1);
        return var == 0 ? 1.0 : std::sqrt(var);
    }
    void Reset() { runningMean = runningVariance = 0; count = 0; }
};
}
