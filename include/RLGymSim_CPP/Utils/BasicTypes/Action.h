// Action / ActionSet (SIM/Utils/BasicTypes/Action.h:5-76): 8 floats {throttle, steer, pitch, yaw, roll, jump, boost, handbrake}
#pragma once
#include "../../Framework.h"
namespace RLGSC {
struct Action {
    float throttle = 0, steer = 0, pitch = 0, yaw = 0, roll = 0, jump = 0, boost = 0, handbrake = 0;
    constexpr static int ELEM_AMOUNT = 8;
    float& operator[](size_t i) { return (&throttle)[i]; }
    float operator[](size_t i) const { return (&throttle)[i]; }
    FList ToFList() const { return FList(&throttle, &throttle + ELEM_AMOUNT); }
};
typedef std::vector<Action> ActionSet;
}
