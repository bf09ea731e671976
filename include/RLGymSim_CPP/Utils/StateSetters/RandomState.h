// RandomState -- every reset scatters the ball and the cars over the field.
//
// On the reference's host path this class mutates an Arena (SIM/Utils/StateSetters/RandomState.cpp:8-61: kickoff reset first, then a
// uniformly placed ball with optional random velocity, then every car with random position / yaw and optional velocity, on the ground
// or in the air).  Here the object only *describes* that setter: the batched env runs it on the device for all envs at once
// (csrc/arena_gym.h reset_state, counter-based Philox streams per env), and ApplyToDevice() hands the three switches over.
#pragma once
#include "StateSetter.h"

namespace RLGSC {

class RandomState : public StateSetter {
public:
    // the three constructor switches of the reference, same order
    bool randBallSpeed;   // ball gets a random linear / angular velocity
    bool randCarSpeed;    // cars get a random velocity
    bool carsOnGround;    // cars are put on their wheels (z = 17) instead of anywhere up to the ceiling

    RandomState(bool ballSpeed, bool carSpeed, bool onGround)
        : randBallSpeed(ballSpeed), randCarSpeed(carSpeed), carsOnGround(onGround) {}

    bool ApplyToDevice(RlgpuGymConfig& deviceCfg) const override {
        deviceCfg.setter_kind = RLGPU_SS_RANDOM;
        deviceCfg.rand_ball_speed = randBallSpeed ? 1 : 0;
        deviceCfg.rand_car_speed = randCarSpeed ? 1 : 0;
        deviceCfg.cars_on_ground = carsOnGround ? 1 : 0;
        return true;
    }
};

}  // namespace RLGSC
