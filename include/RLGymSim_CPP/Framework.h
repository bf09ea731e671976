// RLGymSim_CPP/Framework.h -- host-side mirror of the reference's sim-layer vocabulary (RLGymSim_CPP/src/RLGymSim_CPP/Framework.h,
// RocketSim/src/Math/MathTypes.h) for programs written against RLGymSim_CPP.  The simulation itself runs on the GPU behind
// include/rlgpu.h; these types are what user code (callbacks, env-creation functions) sees.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <filesystem>
#include <functional>
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
// Car.h:17-123 -- the fields the gym layer and typical callbacks read
struct CarState {
    Vec pos, vel, angVel; RotMat rotMat;
    bool isOnGround = true, hasJumped = false, hasDoubleJumped = false, hasFlipped = false, isJumping = false, isFlipping = false;
    bool isSupersonic = false, isDemoed = false;
    float boost = 33.f, airTimeSinceJump = 0, jumpTime = 0, flipTime = 0, demoRespawnTimer = 0;
    CarControls lastControls;
};
struct BallState { Vec pos{0, 0, 93.15f}, vel, angVel; };
// RocketSim::Init(collision_meshes folder) (RS/RocketSim.cpp:70-212): here it only records where the arena meshes are; the
// batched env loads "<folder>/soccar/*.cmf" (rlgpu_env_load_cmf_dir) or, when the folder is missing, the procedural soccar mesh.
void Init(const std::filesystem::path& collisionMeshesFolder, bool silent = false);
const std::filesystem::path& GetCollisionMeshFolder();
}  // namespace RocketSim

namespace RLGSC {
using namespace RocketSim;  // the reference injects it the same way (SIM/Framework.h:8)
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
}
}  // namespace RLGSC
