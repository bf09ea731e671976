// RandomState -- every reset scatters the ball and the cars over the field.
//
// Two forms.  During training the batched env runs the setter on the device for all envs at once (csrc/arena_gym.h reset_state,
// counter-based Philox streams per env); ApplyToDevice() hands the three switches over.  ResetState(Arena*) is the reference's host
// form on the Arena facade: what a standalone Gym calls, and what a user setter deriving from RandomState can build on.
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

    // Host form (RandomState.cpp:8-61) with the thread's engine (::Math::RandFloat): a kickoff reset for pads and bookkeeping, then the
    // ball anywhere inside |x| <= 3500, |y| <= 4000, z <= 1820 and every car likewise, yawed at random; cars on the ground sit at z = 17 with
    // no pitch / roll / vertical or angular velocity.
    GameState ResetState(Arena* arena) override {
        constexpr float X_MAX = 3500, Y_MAX = 4000, Z_MAX = 1820, CAR_Z_MIN = 150, PI = 3.14159265358979323846f;
        auto unitVec = [] { return Math::RandVec(Vec(-1, -1, -1), Vec(1, 1, 1)).Normalized(); };
        arena->ResetToRandomKickoff();
        BallState ball;
        ball.pos = Math::RandVec(Vec(-X_MAX, -Y_MAX, CommonValues::BALL_RADIUS), Vec(X_MAX, Y_MAX, Z_MAX));
        if (randBallSpeed) {
            const Vec dir = unitVec();
            ball.vel = dir * ::Math::RandFloat(0, 4000);
            ball.angVel = Math::RandVec(Vec(-4, -4, -4), Vec(4, 4, 4));
        }
        arena->ball->SetState(ball);
        for (Car* car : arena->_cars) {
            CarState cs;
            cs.pos = Math::RandVec(Vec(-X_MAX, -Y_MAX, CAR_Z_MIN), Vec(X_MAX, Y_MAX, Z_MAX));
            if (randCarSpeed) {
                (void)unitVec();                                   // the reference draws a direction here that it never uses
                const Vec dir = unitVec();
                cs.vel = dir * ::Math::RandFloat(0, RLConst::CAR_MAX_SPEED);
                cs.angVel = unitVec() * CommonValues::CAR_MAX_ANG_VEL;
            }
            const float yaw = ::Math::RandFloat(-PI, PI); const float pitch = ::Math::RandFloat(-PI / 2, PI / 2); const float roll = ::Math::RandFloat(-PI, PI);
            Angle angle(yaw, pitch, roll);
            const bool grounded = carsOnGround ? true : (::Math::RandFloat() > 0.5f);
            if (grounded) { cs.pos.z = 17; angle.pitch = angle.roll = 0; cs.vel.z = 0; cs.angVel = Vec(); }
            cs.rotMat = angle.ToRotMat();
            cs.boost = ::Math::RandFloat(0, 100);
            car->SetState(cs);
        }
        return GameState(arena);
    }

    bool ApplyToDevice(RlgpuGymConfig& deviceCfg) const override { if (!RLG_IS_EXACTLY(RandomState)) return false;
        deviceCfg.setter_kind = RLGPU_SS_RANDOM;
        deviceCfg.rand_ball_speed = randBallSpeed ? 1 : 0;
        deviceCfg.rand_car_speed = randCarSpeed ? 1 : 0;
        deviceCfg.cars_on_ground = carsOnGround ? 1 : 0;
        return true;
    }
};

}  // namespace RLGSC
