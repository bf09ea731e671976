// RLGymSim_CPP/Framework.h -- host-side mirror of the reference's sim-layer vocabulary (RLGymSim_CPP/src/RLGymSim_CPP/Framework.h,
// RocketSim/src/Math/MathTypes.h) for programs written against RLGymSim_CPP.  The simulation itself runs on the GPU behind
// include/rlgpu.h; these types are what user code (callbacks, env-creation functions) sees.
#pragma once
#include <chrono>
#include <cmath>
#include <cstring>
#include <random>
#include <thread>
#include <typeinfo>
#include <cstdint>
#include <cstddef>
#include <filesystem>
#include <functional>
#include "../rlgpu.h"
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

// fatal-error convention of the reference (SIM/Framework.h:17-22): an exception, never an error code
#define RG_ERR_CLOSE(msg_stream)                                                     \
    do {                                                                             \
        std::ostringstream _rg_err; _rg_err << "RG FATAL ERROR: " << msg_stream;     \
        throw std::runtime_error(_rg_err.str());                                     \
    } while (0)

namespace RocketSim {
// MathTypes.h:11-118 (the subset user code touches)
struct Vec {
    float x = 0, y = 0, z = 0;
    constexpr Vec() = default;
    constexpr Vec(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float LengthSq() const { return x * x + y * y + z * z; }
    float Length() const { return std::sqrt(LengthSq()); }
    float Dot(const Vec& o) const { return x * o.x + y * o.y + z * o.z; }
    float Dist(const Vec& o) const { return (*this - o).Length(); }
    Vec Normalized() const { float l = Length(); return l > 1e-12f ? Vec(x / l, y / l, z / l) : Vec(); }
    Vec operator+(const Vec& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec operator-(const Vec& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec operator*(const Vec& o) const { return {x * o.x, y * o.y, z * o.z}; }
    Vec operator*(float s) const { return {x * s, y * s, z * s}; }
    Vec operator/(float s) const { return {x / s, y / s, z / s}; }
    Vec operator-() const { return {-x, -y, -z}; }
    Vec& operator+=(const Vec& o) { x += o.x; y += o.y; z += o.z; return *this; }
    float& operator[](size_t i) { return (&x)[i]; }
    float operator[](size_t i) const { return (&x)[i]; }
};
struct RotMat {  // columns forward / right / up (MathTypes.h:162)
    Vec forward{1, 0, 0}, right{0, 1, 0}, up{0, 0, 1};
};
enum class Team : uint8_t { BLUE = 0, ORANGE = 1 };
enum class GameMode : uint8_t { SOCCAR = 0 };
struct CarControls { float throttle = 0, steer = 0, pitch = 0, yaw = 0, roll = 0; bool jump = false, boost = false, handbrake = false; };

// RLConst.h -- the constants user code reads (values of the game; the simulation's own copies live in csrc/)
namespace RLConst {
constexpr float GRAVITY_Z = -650.f, CAR_MASS_BT = 180.f, BALL_MASS_BT = CAR_MASS_BT / 6.f, BALL_REST_Z = 93.15f, BALL_DRAG = 0.03f;
constexpr float CAR_MAX_SPEED = 2300.f, BALL_MAX_SPEED = 6000.f, BOOST_MAX = 100.f, BOOST_USED_PER_SECOND = BOOST_MAX / 3, BOOST_SPAWN_AMOUNT = BOOST_MAX / 3;
constexpr float DOUBLEJUMP_MAX_DELAY = 1.25f, BALL_COLLISION_RADIUS_SOCCAR = 91.25f, SOCCAR_GOAL_SCORE_BASE_THRESHOLD_Y = 5124.25f;
constexpr float CAR_SPAWN_REST_Z = 17.f, CAR_RESPAWN_Z = 36.f, BUMP_COOLDOWN_TIME = 0.25f, DEMO_RESPAWN_TIME = 3.f;
constexpr float CARWORLD_COLLISION_FRICTION = 0.3f, CARWORLD_COLLISION_RESTITUTION = 0.3f, BALL_FRICTION = 0.35f, BALL_RESTITUTION = 0.6f;
constexpr float BOOST_ACCEL_GROUND = 2975 / 3.f, BOOST_ACCEL_AIR = 3175 / 3.f, JUMP_ACCEL = 4375.f / 3.f, JUMP_IMMEDIATE_FORCE = 875.f / 3.f;
struct CarSpawnPos { float x, y, yawAng; };
constexpr int CAR_SPAWN_LOCATION_AMOUNT = 5, CAR_RESPAWN_LOCATION_AMOUNT = 4;
constexpr float QUARTER_PI = 0.78539816339744830962f;
constexpr CarSpawnPos CAR_SPAWN_LOCATIONS_SOCCAR[CAR_SPAWN_LOCATION_AMOUNT] = {   // two diagonal, two off-centre, one goalie spot (blue side)
    {-2048, -2560, QUARTER_PI * 1}, {2048, -2560, QUARTER_PI * 3}, {-256, -3840, QUARTER_PI * 2}, {256, -3840, QUARTER_PI * 2}, {0, -4608, QUARTER_PI * 2}};
constexpr CarSpawnPos CAR_RESPAWN_LOCATIONS_SOCCAR[CAR_RESPAWN_LOCATION_AMOUNT] = {
    {-2304, -4608, QUARTER_PI * 2}, {-2688, -4608, QUARTER_PI * 2}, {2304, -4608, QUARTER_PI * 2}, {2688, -4608, QUARTER_PI * 2}};
namespace BoostPads { constexpr int LOCS_AMOUNT_BIG = 6, LOCS_AMOUNT_SMALL_SOCCAR = 28; constexpr float COOLDOWN_BIG = 10, COOLDOWN_SMALL = 4, BOOST_AMOUNT_BIG = 100, BOOST_AMOUNT_SMALL = 12; }
}  // namespace RLConst

// Math.h: the thread's random engine behind the host state setters (Math.cpp:44-64).  SeedRandEngine is an addition: the reference seeds
// from the clock only.
namespace Math {
inline std::default_random_engine& GetRandEngine() {
    static thread_local std::default_random_engine engine((std::default_random_engine::result_type)(
        std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count() + std::hash<std::thread::id>()(std::this_thread::get_id())));
    return engine;
}
inline void SeedRandEngine(uint64_t seed) { GetRandEngine().seed((std::default_random_engine::result_type)seed); }
inline int RandInt(int min, int max, int seed = -1) {
    if (seed != -1) { std::default_random_engine tmp((std::default_random_engine::result_type)seed); return min + (int)(tmp() % (unsigned)(max - min)); }
    return min + (int)(GetRandEngine()() % (unsigned)(max - min));
}
inline float RandFloat(float min = 0, float max = 1) {
    std::default_random_engine& e = GetRandEngine();
    return min + ((e() / (float)e.max()) * (max - min));
}
}  // namespace Math

// yaw / pitch / roll (MathTypes.h:281-320); ToRotMat = btMatrix3x3::setEulerYPR(yaw, -pitch, -roll) transposed into forward/right/up columns
struct Angle {
    float yaw, pitch, roll;
    Angle(float yaw = 0, float pitch = 0, float roll = 0) : yaw(yaw), pitch(pitch), roll(roll) {}
    RotMat ToRotMat() const {
        const float ez = yaw, ey = -pitch, ex = -roll;
        const float ci = std::cos(ex), cj = std::cos(ey), ch = std::cos(ez), si = std::sin(ex), sj = std::sin(ey), sh = std::sin(ez);
        const float cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
        RotMat m;   // rows of the bullet matrix are (fwd.x right.x up.x), ...
        m.forward = Vec(cj * ch, cj * sh, -sj); m.right = Vec(sj * sc - cs, sj * ss + cc, cj * si); m.up = Vec(sj * cc + ss, sj * cs - sc, cj * ci);
        return m;
    }
    Vec GetForwardVec() const { return ToRotMat().forward; }
};

struct BallHitInfo {   // BallHitInfo.h:10-25
    bool isValid = false;
    Vec relativePosOnBall, ballPos, extraHitVel;
    uint64_t tickCountWhenHit = ~0ULL, tickCountWhenExtraImpulseApplied = ~0ULL;
};
struct PhysState { Vec pos, vel, angVel; RotMat rotMat; };
// Car.h:17-123, every field
struct CarState : PhysState {
    uint64_t updateCounter = 0;
    bool isOnGround = true; bool wheelsWithContact[4] = {};
    bool hasJumped = false, hasDoubleJumped = false, hasFlipped = false;
    Vec flipRelTorque;
    float jumpTime = 0, flipTime = 0;
    bool isFlipping = false, isJumping = false;
    float airTime = 0, airTimeSinceJump = 0;
    float boost = RLConst::BOOST_SPAWN_AMOUNT, timeSpentBoosting = 0;
    bool isSupersonic = false; float supersonicTime = 0, handbrakeVal = 0;
    bool isAutoFlipping = false; float autoFlipTimer = 0, autoFlipTorqueScale = 0;
    struct { bool hasContact = false; Vec contactNormal; } worldContact;
    struct { uint32_t otherCarID = 0; float cooldownTimer = 0; } carContact;
    bool isDemoed = false; float demoRespawnTimer = 0;
    BallHitInfo ballHitInfo;
    CarControls lastControls;
    CarState() { pos.z = RLConst::CAR_SPAWN_REST_Z; }
    bool HasFlipOrJump() const { return isOnGround || (!hasFlipped && !hasDoubleJumped && airTimeSinceJump < RLConst::DOUBLEJUMP_MAX_DELAY); }   // Car.cpp:285-297
    bool HasFlipReset() const { return !isOnGround && HasFlipOrJump() && !hasJumped; }
    bool GotFlipReset() const { return !isOnGround && !hasJumped; }
};
struct BallState : PhysState {   // Ball.h:17-44
    uint64_t updateCounter = 0;
    BallState() { pos.z = RLConst::BALL_REST_Z; }
};
struct BoostPadConfig { Vec pos; bool isBig = false; };
struct BoostPadState { bool isActive = true; float cooldown = 0; class Car* curLockedCar = nullptr; uint32_t prevLockedCarID = 0; };
// CarConfig.h: the device stepper has the Octane compiled in (csrc/arena_car.h); the struct exists so that Gym's and AddCar's signatures do
struct WheelPairConfig { float wheelRadius = 0, suspensionRestLength = 0; Vec connectionPointOffset; };
struct CarConfig {
    Vec hitboxSize, hitboxPosOffset; WheelPairConfig frontWheels, backWheels; float dodgeDeadzone = 0.5f;
    int preset = 0;   // 0 = Octane
    bool operator==(const CarConfig& o) const { return preset == o.preset && dodgeDeadzone == o.dodgeDeadzone; }
};
inline const CarConfig CAR_CONFIG_OCTANE = [] { CarConfig c; c.hitboxSize = Vec(120.507f, 86.6994f, 38.6591f); c.hitboxPosOffset = Vec(13.87566f, 0, 20.755f); return c; }();
enum class DemoMode : uint8_t { NORMAL, ON_CONTACT, DISABLED };
// MutatorConfig.h (RS/Sim/MutatorConfig/MutatorConfig.h:18-75), the reference's fields under the reference's names.  The device stepper takes the ones that change
// no collision shape and no mass at run time (RlgpuMutators, include/rlgpu_state.h: 24 of the 27); carMass, ballMass and ballRadius stay compiled in and
// are refused off their defaults where they would be applied (Arena::SetMutatorConfig).
struct MutatorConfig {
    Vec gravity = Vec(0, 0, RLConst::GRAVITY_Z);
    float carMass = RLConst::CAR_MASS_BT, carWorldFriction = RLConst::CARWORLD_COLLISION_FRICTION, carWorldRestitution = RLConst::CARWORLD_COLLISION_RESTITUTION;
    float ballMass = RLConst::BALL_MASS_BT, ballMaxSpeed = RLConst::BALL_MAX_SPEED, ballDrag = RLConst::BALL_DRAG;
    float ballWorldFriction = RLConst::BALL_FRICTION, ballWorldRestitution = RLConst::BALL_RESTITUTION;
    float jumpAccel = RLConst::JUMP_ACCEL, jumpImmediateForce = RLConst::JUMP_IMMEDIATE_FORCE;
    float boostAccelGround = RLConst::BOOST_ACCEL_GROUND, boostAccelAir = RLConst::BOOST_ACCEL_AIR, boostUsedPerSecond = RLConst::BOOST_USED_PER_SECOND;
    float respawnDelay = RLConst::DEMO_RESPAWN_TIME, bumpCooldownTime = RLConst::BUMP_COOLDOWN_TIME;
    float boostPadCooldown_Big = RLConst::BoostPads::COOLDOWN_BIG, boostPadCooldown_Small = RLConst::BoostPads::COOLDOWN_SMALL, carSpawnBoostAmount = RLConst::BOOST_SPAWN_AMOUNT;
    float ballHitExtraForceScale = 1, bumpForceScale = 1, ballRadius = RLConst::BALL_COLLISION_RADIUS_SOCCAR;
    bool unlimitedFlips = false, unlimitedDoubleJumps = false; DemoMode demoMode = DemoMode::NORMAL; bool enableTeamDemos = false;
    float goalBaseThresholdY = RLConst::SOCCAR_GOAL_SCORE_BASE_THRESHOLD_Y;
    MutatorConfig(GameMode = GameMode::SOCCAR) {}
    // the fields the stepper has compiled in are at their defaults
    bool CompiledInFieldsAreDefault() const {
        const MutatorConfig d;
        return carMass == d.carMass && ballMass == d.ballMass && ballRadius == d.ballRadius;
    }
    bool IsDefault() const {
        const MutatorConfig d;
        return CompiledInFieldsAreDefault() && gravity.x == 0 && gravity.y == 0 && gravity.z == d.gravity.z && carWorldFriction == d.carWorldFriction &&
               carWorldRestitution == d.carWorldRestitution && ballWorldFriction == d.ballWorldFriction && ballWorldRestitution == d.ballWorldRestitution && ballMaxSpeed == d.ballMaxSpeed && ballDrag == d.ballDrag && jumpAccel == d.jumpAccel &&
               jumpImmediateForce == d.jumpImmediateForce && boostAccelGround == d.boostAccelGround && boostAccelAir == d.boostAccelAir &&
               boostUsedPerSecond == d.boostUsedPerSecond && respawnDelay == d.respawnDelay && bumpCooldownTime == d.bumpCooldownTime &&
               boostPadCooldown_Big == d.boostPadCooldown_Big && boostPadCooldown_Small == d.boostPadCooldown_Small && carSpawnBoostAmount == d.carSpawnBoostAmount &&
               ballHitExtraForceScale == 1 && bumpForceScale == 1 && !unlimitedFlips && !unlimitedDoubleJumps && demoMode == DemoMode::NORMAL &&
               !enableTeamDemos && goalBaseThresholdY == d.goalBaseThresholdY;
    }
    // the run-time fields as the C-ABI takes them (rlgpu_env_set_mutators, RlgpuArenaState::mutators)
    RlgpuMutators ToDevice() const {
        RlgpuMutators m;
        m.gravity_z = gravity.z; m.boost_accel_ground = boostAccelGround; m.boost_accel_air = boostAccelAir; m.boost_used_per_second = boostUsedPerSecond;
        m.jump_accel = jumpAccel; m.jump_immediate_force = jumpImmediateForce; m.ball_max_speed = ballMaxSpeed; m.ball_damp_per_tick = rlgpu_ball_damp_per_tick(ballDrag);
        m.respawn_delay = respawnDelay; m.bump_cooldown_time = bumpCooldownTime; m.boost_pad_cooldown_big = boostPadCooldown_Big; m.boost_pad_cooldown_small = boostPadCooldown_Small;
        m.car_spawn_boost_amount = carSpawnBoostAmount; m.ball_hit_extra_force_scale = ballHitExtraForceScale; m.bump_force_scale = bumpForceScale; m.goal_base_threshold_y = goalBaseThresholdY;
        m.flags = (unlimitedFlips ? RLGPU_MUT_UNLIMITED_FLIPS : 0u) | (unlimitedDoubleJumps ? RLGPU_MUT_UNLIMITED_DOUBLE_JUMPS : 0u) |
                  (demoMode == DemoMode::ON_CONTACT ? RLGPU_MUT_DEMO_ON_CONTACT : demoMode == DemoMode::DISABLED ? RLGPU_MUT_DEMO_DISABLED : 0u) | (enableTeamDemos ? RLGPU_MUT_TEAM_DEMOS : 0u);
        m._pad = 0;
        m.gravity_x = gravity.x; m.gravity_y = gravity.y; m.car_world_friction = carWorldFriction; m.car_world_restitution = carWorldRestitution;
        m.ball_world_friction = ballWorldFriction; m.ball_world_restitution = ballWorldRestitution;
        return m;
    }
};
struct ArenaConfig {};
// RocketSim::Init(collision_meshes folder) (RS/RocketSim.cpp:70-212): here it only records where the arena meshes are; the
// batched env loads "<folder>/soccar/*.cmf" (rlgpu_env_load_cmf_dir) or, when the folder is missing, the procedural soccar mesh.
void Init(const std::filesystem::path& collisionMeshesFolder, bool silent = false);
const std::filesystem::path& GetCollisionMeshFolder();
}  // namespace RocketSim

using namespace RocketSim;  // the reference injects it the same way, at global scope (SIM/Framework.h:8) -- `::Math::RandFloat` in user setters

// A built-in plugin's device form is valid for objects of exactly that class: a user subclass may override any host-side virtual
// (BuildOBS, AddPlayerToOBS, GetReward, ...), which the step kernel would not see -- such an object runs on the host instead.
#define RLG_IS_EXACTLY(Class) (typeid(*this) == typeid(Class))

namespace RLGSC {
using namespace RocketSim;  // so that RLGSC::Vec, RLGSC::Arena ... name them too
typedef std::vector<float> FList;
typedef std::vector<FList> FList2;
typedef std::vector<int> IList;
inline FList& operator+=(FList& l, float v) { l.push_back(v); return l; }
inline FList& operator+=(FList& l, const Vec& v) { l.push_back(v.x); l.push_back(v.y); l.push_back(v.z); return l; }
inline FList& operator+=(FList& l, const FList& o) { l.insert(l.end(), o.begin(), o.end()); return l; }
inline FList& operator+=(FList& l, std::initializer_list<float> o) { l.insert(l.end(), o.begin(), o.end()); return l; }

namespace CommonValues {  // Utils/CommonValues.h
constexpr float SIDE_WALL_X = 4096, BACK_WALL_Y = 5120, CEILING_Z = 2044, BACK_NET_Y = 6000, GOAL_HEIGHT = 642.775f;
constexpr float BALL_RADIUS = 92.75f, BALL_MAX_SPEED = 6000, CAR_MAX_SPEED = 2300, SUPERSONIC_THRESHOLD = 2200, CAR_MAX_ANG_VEL = 5.5f;
constexpr int BLUE_TEAM = 0, ORANGE_TEAM = 1, NUM_ACTIONS = 8, BOOST_LOCATIONS_AMOUNT = 34;
constexpr Vec BLUE_GOAL_BACK = Vec(0, -BACK_NET_Y, GOAL_HEIGHT / 2), ORANGE_GOAL_BACK = Vec(0, BACK_NET_Y, GOAL_HEIGHT / 2);
constexpr Vec BLUE_GOAL_CENTER = Vec(0, -BACK_WALL_Y, GOAL_HEIGHT / 2), ORANGE_GOAL_CENTER = Vec(0, BACK_WALL_Y, GOAL_HEIGHT / 2);
}
namespace Math {  // SIM/Math.h
inline bool IsBallScored(Vec pos) { return std::fabs(pos.y) > RLConst::SOCCAR_GOAL_SCORE_BASE_THRESHOLD_Y + RLConst::BALL_COLLISION_RADIUS_SOCCAR; }
inline Vec RandVec(Vec min, Vec max) {   // three draws, x then y then z (Math.cpp:7-13)
    const float x = ::Math::RandFloat(min.x, max.x); const float y = ::Math::RandFloat(min.y, max.y); const float z = ::Math::RandFloat(min.z, max.z);
    return Vec(x, y, z);
}
}
}  // namespace RLGSC
