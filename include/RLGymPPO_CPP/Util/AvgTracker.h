// AvgTracker -- a running mean for metrics (PUB/Util/AvgTracker.h:5-50).
//
// Behaviour user callbacks rely on (the example's OnIteration adds per-game averages with +=):
//   * NaN samples are ignored, so an empty game report does not poison the mean;
//   * Get() of an empty tracker is NaN (not 0), which the metrics code prints as "nan";
//   * `a += b` with another tracker merges totals and counts (a weighted mean), `a += x` adds one sample.
#pragma once
#include "../Framework.h"

namespace RLGPC {

struct AvgTracker {
    float total = 0;
    uint64_t count = 0;

    float Get() const {
        if (count == 0) return NAN;
        return total / count;
    }

    void Add(float sample) {
        if (std::isnan(sample)) return;
        total += sample;
        count += 1;
    }
    // `n` samples whose sum is `sum`
    void Add(float sum, uint64_t n) {
        if (std::isnan(sum)) return;
        total += sum;
        count += n;
    }

    AvgTracker& operator+=(float sample) { Add(sample); return *this; }
    AvgTracker& operator+=(const AvgTracker& other) { Add(other.total, other.count); return *this; }

    void Reset() { total = 0; count = 0; }
};

}  // namespace RLGPC
