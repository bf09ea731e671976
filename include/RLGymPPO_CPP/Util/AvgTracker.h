// AvgTracker (PUB/Util/AvgTracker.h): running mean that ignores NaN samples; Get() is NaN while empty
#pragma once
#include "../Framework.h"
namespace RLGPC {
struct AvgTracker {
    float total = 0; uint64_t count = 0;
    float Get() const { return count ? total / count : NAN; }
    void Add(float v) { if (!std::isnan(v)) { total += v; count++; } }
    void Add(float totalVal, uint64_t n) { if (!std::isnan(totalVal)) { total += totalVal; count += n; } }
    AvgTracker& operator+=(float v) { Add(v); return *this; }
    AvgTracker& operator+=(const AvgTracker& o) { Add(o.total, o.count); return *this; }
    void Reset() { total = 0; count = 0; }
};
}
