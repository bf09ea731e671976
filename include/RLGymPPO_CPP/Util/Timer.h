#pragma once
#include "../Framework.h"
namespace RLGPC {
struct Timer {
    std::chrono::steady_clock::time_point start = std::chrono::steady_clock::now();
    double Elapsed() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count(); }
    void Reset() { start = std::chrono::steady_clock::now(); }
};
}
