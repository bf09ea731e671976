// Arena / Car / Ball / BoostPad -- the objects user plugins written against the reference reach through `Arena*`
// (RS/Sim/Arena/Arena.h:26-220, RS/Sim/Car/Car.h:125-236, RS/Sim/Ball/Ball.h:47-96, RS/Sim/BoostPad/BoostPad.h:33-58).
//
// In this build an arena lives on the GPU as one column of the batched env.  What a StateSetter, a Gym or a
// test gets is this HOST FACADE: an Arena owns one RlgpuArenaState (the exchange layout of rlgpu_state.h) and Car / Ball / BoostPad are
// views into it with the reference's GetState / SetState semantics.  The batched env hands such an arena to a user StateSetter,
// uploads what it left there (rlgpu_env_upload_states) and goes on stepping on the device; a standalone Gym steps its arena through
// a one-env device batch (Arena::Step, host/Gym.hip).  Nothing here simulates on the CPU.
#pragma once
#include <algorithm>
#include <array>
#include <random>
#include "../Framework.h"
#include "../../rlgpu.h"

namespace RocketSim {

// ---- CarState / BallState <-> exchange layout -----------------------------------------------------------------------------------
inline void CarStateToRlgpu(const CarState& s, RlgpuCarState& c) {   // what Car::SetState leaves in the car (Car.cpp:23-36)
    auto put = [](float* d, const Vec& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
    put(c.pos, s.pos); put(c.rot, s.rotMat.forward); put(c.rot + 3, s.rotMat.right); put(c.rot + 6, s.rotMat.up); put(c.vel, s.vel); put(c.ang_vel, s.angVel);
    uint32_t f = 0;
    if (s.isOnGround) f |= RLGPU_CF_ON_GROUND;
    for (int w = 0; w < 4; w++) if (s.wheelsWithContact[w]) f |= (RLGPU_CF_WHEEL0 << w);
    if (s.hasJumped) f |= RLGPU_CF_HAS_JUMPED;
    if (s.hasDoubleJumped) f |= RLGPU_CF_HAS_DOUBLE_JUMPED;
    if (s.hasFlipped) f |= RLGPU_CF_HAS_FLIPPED;
    if (s.isFlipping) f |= RLGPU_CF_IS_FLIPPING;
    if (s.isJumping) f |= RLGPU_CF_IS_JUMPING;
    if (s.isSupersonic) f |= RLGPU_CF_IS_SUPERSONIC;
    if (s.isAutoFlipping) f |= RLGPU_CF_IS_AUTOFLIPPING;
    if (s.worldContact.hasContact) f |= RLGPU_CF_WORLD_CONTACT;
    if (s.isDemoed) f |= RLGPU_CF_IS_DEMOED;
    if (s.ballHitInfo.isValid) f |= RLGPU_CF_BALLHIT_VALID;
    c.flags = f;
    put(c.flip_rel_torque, s.flipRelTorque);
    c.jump_time = s.jumpTime; c.flip_time = s.flipTime; c.air_time = s.airTime; c.air_time_since_jump = s.airTimeSinceJump;
    c.boost = s.boost; c.time_spent_boosting = s.timeSpentBoosting; c.supersonic_time = s.supersonicTime; c.handbrake_val = s.handbrakeVal;
    c.auto_flip_timer = s.autoFlipTimer; c.auto_flip_torque_scale = s.autoFlipTorqueScale;
    put(c.world_contact_normal, s.worldContact.contactNormal);
    c.car_contact_other_id = (int32_t)s.carContact.otherCarID; c.car_contact_cooldown = s.carContact.cooldownTimer;
    c.demo_respawn_timer = s.demoRespawnTimer;
    put(c.bh_rel_pos, s.ballHitInfo.relativePosOnBall); put(c.bh_ball_pos, s.ballHitInfo.ballPos); put(c.bh_extra_hit_vel, s.ballHitInfo.extraHitVel);
    c.bh_tick_hit = s.ballHitInfo.tickCountWhenHit == ~0ULL ? -1 : (int64_t)s.ballHitInfo.tickCountWhenHit;
    c.bh_tick_extra = s.ballHitInfo.tickCountWhenExtraImpulseApplied == ~0ULL ? -1 : (int64_t)s.ballHitInfo.tickCountWhenExtraImpulseApplied;
    const CarControls& l = s.lastControls;
    const float lc[8] = {l.throttle, l.steer, l.pitch, l.yaw, l.roll, (float)l.jump, (float)l.boost, (float)l.handbrake};
    for (int i = 0; i < 8; i++) c.last_controls[i] = lc[i];
    c.vel_impulse_cache[0] = c.vel_impulse_cache[1] = c.vel_impulse_cache[2] = 0.f;   // Car.cpp:32
    // the wheels' carried values (extra_pushback, wheel_*) are not part of CarState and survive, like the reference's btWheelInfoRL
}
inline CarState CarStateFromRlgpu(const RlgpuCarState& c) {   // Car::GetState (Car.cpp:10-20)
    auto V = [](const float* p) { return Vec(p[0], p[1], p[2]); };
    CarState s;
    s.pos = V(c.pos); s.rotMat.forward = V(c.rot); s.rotMat.right = V(c.rot + 3); s.rotMat.up = V(c.rot + 6); s.vel = V(c.vel); s.angVel = V(c.ang_vel);
    const uint32_t f = c.flags;
    s.isOnGround = f & RLGPU_CF_ON_GROUND;
    for (int w = 0; w < 4; w++) s.wheelsWithContact[w] = f & (RLGPU_CF_WHEEL0 << w);
    s.hasJumped = f & RLGPU_CF_HAS_JUMPED; s.hasDoubleJumped = f & RLGPU_CF_HAS_DOUBLE_JUMPED; s.hasFlipped = f & RLGPU_CF_HAS_FLIPPED;
    s.isFlipping = f & RLGPU_CF_IS_FLIPPING; s.isJumping = f & RLGPU_CF_IS_JUMPING; s.isSupersonic = f & RLGPU_CF_IS_SUPERSONIC;
    s.isAutoFlipping = f & RLGPU_CF_IS_AUTOFLIPPING; s.worldContact.hasContact = f & RLGPU_CF_WORLD_CONTACT; s.isDemoed = f & RLGPU_CF_IS_DEMOED;
    s.flipRelTorque = V(c.flip_rel_torque);
    s.jumpTime = c.jump_time; s.flipTime = c.flip_time; s.airTime = c.air_time; s.airTimeSinceJump = c.air_time_since_jump;
    s.boost = c.boost; s.timeSpentBoosting = c.time_spent_boosting; s.supersonicTime = c.supersonic_time; s.handbrakeVal = c.handbrake_val;
    s.autoFlipTimer = c.auto_flip_timer; s.autoFlipTorqueScale = c.auto_flip_torque_scale;
    s.worldContact.contactNormal = V(c.world_contact_normal);
    s.carContact.otherCarID = (uint32_t)c.car_contact_other_id; s.carContact.cooldownTimer = c.car_contact_cooldown;
    s.demoRespawnTimer = c.demo_respawn_timer;
    s.ballHitInfo.isValid = f & RLGPU_CF_BALLHIT_VALID;
    s.ballHitInfo.relativePosOnBall = V(c.bh_rel_pos); s.ballHitInfo.ballPos = V(c.bh_ball_pos); s.ballHitInfo.extraHitVel = V(c.bh_extra_hit_vel);
    s.ballHitInfo.tickCountWhenHit = c.bh_tick_hit < 0 ? ~0ULL : (uint64_t)c.bh_tick_hit;
    s.ballHitInfo.tickCountWhenExtraImpulseApplied = c.bh_tick_extra < 0 ? ~0ULL : (uint64_t)c.bh_tick_extra;
    CarControls& l = s.lastControls;
    l.throttle = c.last_controls[0]; l.steer = c.last_controls[1]; l.pitch = c.last_controls[2]; l.yaw = c.last_controls[3]; l.roll = c.last_controls[4];
    l.jump = c.last_controls[5] != 0.f; l.boost = c.last_controls[6] != 0.f; l.handbrake = c.last_controls[7] != 0.f;
    return s;
}

class Arena;

class Ball {
public:
    BallState GetState() const {
        BallState s; const RlgpuBallState& b = *raw;
        s.pos = Vec(b.pos[0], b.pos[1], b.pos[2]); s.vel = Vec(b.vel[0], b.vel[1], b.vel[2]); s.angVel = Vec(b.ang_vel[0], b.ang_vel[1], b.ang_vel[2]);
        s.updateCounter = (uint64_t)*updateCounter;
        // BallState::rotMat (Ball.cpp:27-30): with ArenaConfig::noBallRot the basis the last SetState gave the ball
        s.rotMat.forward = Vec(rawRot[0], rawRot[1], rawRot[2]); s.rotMat.right = Vec(rawRot[3], rawRot[4], rawRot[5]); s.rotMat.up = Vec(rawRot[6], rawRot[7], rawRot[8]);
        return s;
    }
    void SetState(const BallState& s) {   // Ball.cpp:27-49: velocities, the impulse cache and the update counter start over
        RlgpuBallState& b = *raw;
        for (int i = 0; i < 3; i++) { b.pos[i] = s.pos[i]; b.vel[i] = s.vel[i]; b.ang_vel[i] = s.angVel[i]; b.vel_impulse_cache[i] = 0.f; }
        for (int i = 0; i < 3; i++) { rawRot[i] = s.rotMat.forward[i]; rawRot[3 + i] = s.rotMat.right[i]; rawRot[6 + i] = s.rotMat.up[i]; }   // newTransform.setBasis(state.rotMat), Ball.cpp:41
        *updateCounter = 0;
    }
    float GetRadius() const { return RLConst::BALL_COLLISION_RADIUS_SOCCAR; }
private:
    friend class Arena;
    RlgpuBallState* raw = nullptr; int64_t* updateCounter = nullptr; float* rawRot = nullptr;   // (rawRot: RlgpuArenaState::hidden.ball_rot)
};

class Car {
public:
    CarConfig config;
    Team team = Team::BLUE;
    uint32_t id = 0;                       // slot + 1 (rlgpu_state.h)
    CarControls controls;                  // what the next tick uses; written into the state when the arena goes to the device

    CarState GetState() const { return CarStateFromRlgpu(*raw); }
    void SetState(const CarState& s) { CarStateToRlgpu(s, *raw); }
    void Demolish(float respawnDelay = RLConst::DEMO_RESPAWN_TIME) { raw->flags |= RLGPU_CF_IS_DEMOED; raw->demo_respawn_timer = respawnDelay; }   // Car.cpp:38-41
    // Car::Respawn (Car.cpp:43-56): a fresh state on one of the four respawn spots
    void Respawn(GameMode = GameMode::SOCCAR, int seed = -1, float boostAmount = RLConst::BOOST_SPAWN_AMOUNT) {
        CarState ns;
        const RLConst::CarSpawnPos& sp = RLConst::CAR_RESPAWN_LOCATIONS_SOCCAR[Math::RandInt(0, RLConst::CAR_RESPAWN_LOCATION_AMOUNT, seed)];
        const bool blue = team == Team::BLUE;
        ns.pos = Vec(sp.x, sp.y * (blue ? 1.f : -1.f), RLConst::CAR_RESPAWN_Z);
        ns.rotMat = Angle(sp.yawAng + (blue ? 0.f : (float)M_PI), 0.f, 0.f).ToRotMat();
        ns.boost = boostAmount;
        SetState(ns);
    }
    Vec GetForwardDir() const { return Vec(raw->rot[0], raw->rot[1], raw->rot[2]); }
    Vec GetRightDir() const { return Vec(raw->rot[3], raw->rot[4], raw->rot[5]); }
    Vec GetUpDir() const { return Vec(raw->rot[6], raw->rot[7], raw->rot[8]); }
private:
    friend class Arena;
    RlgpuCarState* raw = nullptr;
};

class BoostPad {
public:
    BoostPadConfig config;
    BoostPadState GetState() const {
        BoostPadState s; s.isActive = raw->is_active != 0; s.cooldown = raw->cooldown; s.prevLockedCarID = (uint32_t)raw->prev_locked_car_id;
        return s;
    }
    void SetState(const BoostPadState& s) { raw->is_active = s.isActive ? 1 : 0; raw->cooldown = s.cooldown; raw->prev_locked_car_id = (int32_t)s.prevLockedCarID; }
private:
    friend class Arena;
    RlgpuPadState* raw = nullptr;
};

typedef std::function<void(Arena* arena, Team scoringTeam, void* userInfo)> GoalScoreEventFn;
typedef std::function<void(Arena* arena, Car* bumper, Car* victim, bool isDemo, void* userInfo)> CarBumpEventFn;

class Arena {
public:
    GameMode gameMode = GameMode::SOCCAR;
    uint32_t _lastCarID = 0;
    std::vector<Car*> _cars;               // slot order = id order (the reference keeps a set; every use iterates it)
    Ball* ball;
    std::vector<BoostPad*> _boostPads;     // RocketSim order: 6 big, then 28 small
    float tickTime = 1 / 120.f;
    uint64_t tickCount = 0;
    RlgpuArenaState _state;                // the arena itself, in the exchange layout

    static Arena* Create(GameMode gameMode = GameMode::SOCCAR, const ArenaConfig& = {}, float tickRate = 120) {
        if (gameMode != GameMode::SOCCAR) RG_ERR_CLOSE("Arena::Create(): only GameMode::SOCCAR exists in this build");
        if (tickRate != 120) RG_ERR_CLOSE("Arena::Create(): the device stepper runs at 120 ticks per second");
        return new Arena();
    }
    Arena(const Arena&) = delete;
    Arena& operator=(const Arena&) = delete;
    ~Arena() { ReleaseDevice(); for (Car* c : _cars) delete c; for (BoostPad* p : _boostPads) delete p; delete ball; }

    float GetTickRate() const { return 1 / tickTime; }
    const std::vector<Car*>& GetCars() { return _cars; }
    const std::vector<BoostPad*>& GetBoostPads() { return _boostPads; }
    const MutatorConfig& GetMutatorConfig() { return _mutatorConfig; }
    // Arena::SetMutatorConfig (Arena.cpp:15-48).  The run-time fields travel with the arena's state from here on (RlgpuArenaState::mutators: the next upload
    // hands them to the env); car / ball mass and the ball's radius are compiled into the stepper.
    void SetMutatorConfig(const MutatorConfig& m) {
        if (!m.CompiledInFieldsAreDefault())
            RG_ERR_CLOSE("Arena::SetMutatorConfig(): carMass, ballMass and ballRadius are compiled into the device stepper (defaults only)");
        _mutatorConfig = m;
        _state.mutators = m.ToDevice(); _state.hidden.valid |= RLGPU_HIDDEN_MUTATORS;
    }

    // Arena::AddCar (Arena.cpp:33-69).  The device layout fixes the slots (even = blue, odd = orange), so the cars come in the order Gym's
    // constructor adds them (Gym.cpp:45-49): blue, orange, blue, orange, ... (car id = slot + 1), or -- spawnOpponents = false -- blue cars
    // only (ids 1, 2, 3 on slots 0, 2, 4; the orange slots stay empty, flagged RLGPU_CF_ABSENT).
    Car* AddCar(Team team, const CarConfig& config = CAR_CONFIG_OCTANE) {
        const int n = (int)_cars.size();
        if (!(config == CAR_CONFIG_OCTANE)) RG_ERR_CLOSE("Arena::AddCar(): the device stepper has the Octane hitbox compiled in");
        if (n == 1 && team == Team::BLUE) _oneTeam = true;         // the second car decides: blue again = an arena without opponents (car 0 sits on slot 0 either way)
        const bool oneTeam = _oneTeam;
        if (oneTeam ? team != Team::BLUE : ((n % 2 == 0) != (team == Team::BLUE)))
            RG_ERR_CLOSE("Arena::AddCar(): cars have to be added blue, orange, blue, orange, ... or all blue (the device's slot order)");
        const int slot = oneTeam ? 2 * n : n;
        if (slot >= RLGPU_MAX_CARS) RG_ERR_CLOSE("Arena::AddCar(): at most " << RLGPU_MAX_CARS / (oneTeam ? 2 : 1) << " cars per arena");
        Car* car = new Car();
        car->config = config; car->team = team; car->id = ++_lastCarID; car->raw = &_state.cars[slot];
        _cars.push_back(car);
        _state.num_cars = (int32_t)(oneTeam ? 2 * _cars.size() : _cars.size());
        CarState fresh; fresh.pos = Vec(0, team == Team::BLUE ? -1000.f - 300.f * (slot / 2) : 1000.f + 300.f * (slot / 2), RLConst::CAR_SPAWN_REST_Z);
        car->SetState(fresh);
        if (oneTeam) _MarkAbsentSlots();
        return car;
    }
    // a single blue car is a one-team arena until an orange one joins (Gym adds blue first either way)
    void _MarkAbsentSlots() {
        for (int k = 1; k < _state.num_cars; k += 2) {
            RlgpuCarState& c = _state.cars[k];
            std::memset(&c, 0, sizeof(c));
            c.flags = RLGPU_CF_IS_DEMOED | RLGPU_CF_ABSENT; c.demo_respawn_timer = 1e30f; c.pos[2] = -10000.f;
            c.rot[0] = c.rot[4] = c.rot[8] = 1.f; c.bh_tick_hit = -1; c.bh_tick_extra = -1;
        }
    }
    // an arena that will never get orange cars (Gym with spawnOpponents = false and teamSize 1 has to say so: nothing else tells)
    void _SetOneTeam() { _oneTeam = true; _state.num_cars = (int32_t)(2 * _cars.size()); _MarkAbsentSlots(); }
    bool _IsOneTeam() const { return _oneTeam; }
    Car* _CarOfSlot(int slot) { for (Car* c : _cars) if (c->raw == &_state.cars[slot]) return c; return nullptr; }
    Car* GetCar(uint32_t id) { return (id >= 1 && id <= _cars.size()) ? _cars[id - 1] : nullptr; }

    void SetGoalScoreCallback(GoalScoreEventFn fn, void* userInfo = nullptr) { _goalScoreCallback = {fn, userInfo}; }
    void SetCarBumpCallback(CarBumpEventFn fn, void* userInfo = nullptr) { _carBumpCallback = {fn, userInfo}; }

    // Arena::ResetToRandomKickoff (Arena.cpp:112-216): the five kickoff spots shuffled with the thread's engine (or a seeded one), blue
    // and orange mirrored, ball at rest in the centre, all pads active
    void ResetToRandomKickoff(int seed = -1) {
        using namespace RLConst;
        std::array<int, CAR_SPAWN_LOCATION_AMOUNT> order;
        for (int i = 0; i < CAR_SPAWN_LOCATION_AMOUNT; i++) order[i] = i;
        std::default_random_engine seeded((unsigned)(seed == -1 ? 0 : seed));
        std::default_random_engine& eng = seed == -1 ? Math::GetRandEngine() : seeded;
        std::shuffle(order.begin(), order.end(), eng);
        int perTeam[2] = {0, 0};
        for (Car* car : _cars) {
            const bool blue = car->team == Team::BLUE;
            const int i = perTeam[blue ? 0 : 1]++;
            CarSpawnPos sp;
            if (i < CAR_SPAWN_LOCATION_AMOUNT) sp = CAR_SPAWN_LOCATIONS_SOCCAR[order[i]];
            else sp = CAR_RESPAWN_LOCATIONS_SOCCAR[(i - CAR_SPAWN_LOCATION_AMOUNT) % CAR_RESPAWN_LOCATION_AMOUNT];
            CarState st; st.pos = Vec(sp.x, sp.y, CAR_SPAWN_REST_Z); st.isOnGround = true;
            Angle ang(sp.yawAng, 0, 0);
            if (!blue) { st.pos = st.pos * Vec(-1, -1, 1); ang.yaw += (float)M_PI; }
            st.rotMat = ang.ToRotMat();
            car->SetState(st);
        }
        ball->SetState(BallState());
        for (BoostPad* p : _boostPads) p->SetState(BoostPadState());
    }

    bool IsBallScored() const { return std::fabs(_state.ball.pos[1]) > _mutatorConfig.goalBaseThresholdY + _mutatorConfig.ballRadius; }   // Arena.cpp:949-957

    // Arena::Step (Arena.cpp:716-812) on a one-env device batch (created on first use; librlgymppo_amd.so).  With a goal / bump callback
    // set the arena is stepped tick by tick and the callbacks are raised from the state each tick leaves (host/Gym.hip); the gym layer
    // does not use them (its counters are kept by the device step, csrc/arena_gym.h).
    void Step(int ticksToSimulate = 1);

    // the cars' `controls` members -> the state (before the arena is uploaded), and the tick counter both ways
    void _SyncToState() {
        for (Car* c : _cars) {
            const CarControls& k = c->controls;
            const float v[8] = {k.throttle, k.steer, k.pitch, k.yaw, k.roll, (float)k.jump, (float)k.boost, (float)k.handbrake};
            for (int i = 0; i < 8; i++) c->raw->controls[i] = v[i];
        }
        _state.tick_count = (int64_t)tickCount;
    }
    void _SyncFromState() { tickCount = (uint64_t)_state.tick_count; }
    void* _device = nullptr;               // the one-env device batch behind Step() / Gym (host/Gym.hip)
    bool _oneTeam = false;                 // blue cars only (spawnOpponents = false)

private:
    Arena() {
        std::memset(&_state, 0, sizeof(_state));
        ball = new Ball(); ball->raw = &_state.ball; ball->updateCounter = &_state.ball_update_counter; ball->rawRot = _state.hidden.ball_rot;
        ball->SetState(BallState());
        for (int p = 0; p < RLGPU_NUM_PADS; p++) {
            BoostPad* pad = new BoostPad();
            float at[3] = {0, 0, 0}; int big = 0;
            rlgpu_pad_location(p, at, &big);   // the stepper's own table (csrc/arena_step.h)
            pad->raw = &_state.pads[p]; pad->config.isBig = big != 0; pad->config.pos = Vec(at[0], at[1], at[2]);
            pad->SetState(BoostPadState());
            _boostPads.push_back(pad);
        }
        _state.gym.last_touch_car_id = -1;
        for (int k = 0; k < RLGPU_MAX_CARS; k++) _state.gym.players[k].prev_action_idx = -1;
    }
    void ReleaseDevice();
    MutatorConfig _mutatorConfig{GameMode::SOCCAR};
    struct { GoalScoreEventFn func; void* userInfo = nullptr; } _goalScoreCallback;
    struct { CarBumpEventFn func; void* userInfo = nullptr; } _carBumpCallback;
};

}  // namespace RocketSim
