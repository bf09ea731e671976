// RandomState (SIM/Utils/StateSetters/RandomState.h:5-16, .cpp:8-61)
#pragma once
#include "StateSetter.h"
namespace RLGSC {
class RandomState : public StateSetter {
public:
    bool randBallSpeed, randCarSpeed, carsOnGround;
    RandomState(bool randBallSpeed, bool randCarSpeed, bool carsOnGround) : randBallSpeed(randBallSpeed), randCarSpeed(randCarSpeed), carsOnGround(carsOnGround) {}
    bool ApplyToDevice(RlgpuGymConfig& cfg) const override {
        cfg.setter_kind = RLGPU_SS_RANDOM; cfg.rand_ball_speed = randBallSpeed; cfg.rand_car_speed = randCarSpeed; cfg.cars_on_ground = carsOnGround;
        return true;
    }
};
}
