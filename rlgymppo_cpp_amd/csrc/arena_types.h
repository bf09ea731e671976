// arena_types.h — constants and per-env working types of the batched arena stepper.
//
// The stepper keeps rigid bodies in Bullet units (1 BT = 50 uu) exactly as the reference's
// btRigidBody does (RocketSim/src/BulletLink.h:12-15) so that the fp32 rounding of every product
// follows the reference; the public exchange layout (include/rlgpu_state.h) is in uu.
//
// Everything here restates numbers/semantics of the reference; each block cites where they come from.
#pragma once
#include <cstddef>
#include "rl_math.h"

namespace rlg {

constexpr float UU2BT = 1.f / 50.f;
constexpr float BT2UU = 50.f;
constexpr float TICK_DT = 1.f / 120.f;  // Arena.cpp:437 tickTime = 1 / tickRate

// ---- RocketSim/src/RLConst.h:9-135 ----------------------------------------------------------------
namespace K {
constexpr float GRAVITY_Z = -650.f;
constexpr float ARENA_EXTENT_X = 4096.f, ARENA_EXTENT_Y = 5120.f, ARENA_HEIGHT = 2048.f;
constexpr float CAR_MASS = 180.f, BALL_MASS = 180.f / 6.f;
constexpr float CARBALL_FRICTION = 2.0f, CARBALL_RESTITUTION = 0.0f;
constexpr float CARWORLD_FRICTION = 0.3f, CARWORLD_RESTITUTION = 0.3f;
constexpr float CARCAR_FRICTION = 0.09f, CARCAR_RESTITUTION = 0.1f;
constexpr float BALL_REST_Z = 93.15f, BALL_MAX_ANG_SPEED = 6.f, BALL_DRAG = 0.03f;
// powf(1 - BALL_DRAG, 1 / 120) and powf(1 - FLIP_Z_DAMP_120, 1) as glibc rounds them (constants, so that no libm call is left in the tick;
// tests/cpp/libm_check.cpp checks both against the C library)
constexpr float BALL_DAMP_PER_TICK = 0x1.ffdebcp-1f, FLIP_Z_DAMP_PER_TICK = 0x1.4cccccp-1f;
constexpr float BALL_FRICTION = 0.35f, BALL_RESTITUTION = 0.6f;  // vs world: min(0.35,0.6), max(0.6,0.3) (btManifoldResult.cpp:56-78)
constexpr float CAR_MAX_SPEED = 2300.f, BALL_MAX_SPEED = 6000.f;
constexpr float BOOST_MAX = 100.f, BOOST_USED_PER_SECOND = BOOST_MAX / 3, BOOST_MIN_TIME = 0.1f;
constexpr float BOOST_ACCEL_GROUND = 2975 / 3.f, BOOST_ACCEL_AIR = 3175 / 3.f, BOOST_SPAWN_AMOUNT = BOOST_MAX / 3;
constexpr float CAR_MAX_ANG_SPEED = 5.5f;
constexpr float SUPERSONIC_START_SPEED = 2200.f, SUPERSONIC_MAINTAIN_MIN_SPEED = SUPERSONIC_START_SPEED - 100.f,
                SUPERSONIC_MAINTAIN_MAX_TIME = 1.f;
constexpr float POWERSLIDE_RISE_RATE = 5, POWERSLIDE_FALL_RATE = 2;
constexpr float THROTTLE_TORQUE_AMOUNT = CAR_MASS * 400.f, BRAKE_TORQUE_AMOUNT = CAR_MASS * (14.25f + (1.f / 3.f));
constexpr float STOPPING_FORWARD_VEL = 25.f, COASTING_BRAKE_FACTOR = 0.15f, BRAKING_NO_THROTTLE_SPEED_THRESH = 0.01f,
                THROTTLE_DEADZONE = 0.001f, THROTTLE_AIR_ACCEL = 200 / 3.f;
constexpr float JUMP_ACCEL = 4375.f / 3.f, JUMP_IMMEDIATE_FORCE = 875.f / 3.f, JUMP_MIN_TIME = 0.025f,
                JUMP_RESET_TIME_PAD = (1 / 40.f), JUMP_MAX_TIME = 0.2f, DOUBLEJUMP_MAX_DELAY = 1.25f;
constexpr float FLIP_Z_DAMP_120 = 0.35f, FLIP_Z_DAMP_START = 0.15f, FLIP_Z_DAMP_END = 0.21f, FLIP_TORQUE_TIME = 0.65f,
                FLIP_PITCHLOCK_EXTRA_TIME = 0.3f, FLIP_INITIAL_VEL_SCALE = 500.f, FLIP_TORQUE_X = 260.f, FLIP_TORQUE_Y = 224.f,
                FLIP_FORWARD_IMPULSE_MAX_SPEED_SCALE = 1.f, FLIP_SIDE_IMPULSE_MAX_SPEED_SCALE = 1.9f,
                FLIP_BACKWARD_IMPULSE_MAX_SPEED_SCALE = 2.5f, FLIP_BACKWARD_IMPULSE_SCALE_X = 16.f / 15.f;
constexpr float BALL_RADIUS = 91.25f;
constexpr float GOAL_THRESHOLD_Y = 5124.25f;
constexpr float CAR_TORQUE_SCALE = (float)(2 * 3.14159265358979323846 / (1 << 16) * 1000);
constexpr float CAR_AUTOFLIP_IMPULSE = 200, CAR_AUTOFLIP_TORQUE = 50, CAR_AUTOFLIP_TIME = 0.4f,
                CAR_AUTOFLIP_NORMZ_THRESH = 0.70710678118654752440f, CAR_AUTOFLIP_ROLL_THRESH = 2.8f;
constexpr float CAR_AUTOROLL_FORCE = 100, CAR_AUTOROLL_TORQUE = 80;
constexpr float BALL_CAR_EXTRA_IMPULSE_Z_SCALE = 0.35f, BALL_CAR_EXTRA_IMPULSE_FORWARD_SCALE = 0.65f,
                BALL_CAR_EXTRA_IMPULSE_MAXDELTAVEL_UU = 4600.f;
constexpr float CAR_SPAWN_REST_Z = 17.f, CAR_RESPAWN_Z = 36.f;
constexpr float BUMP_COOLDOWN_TIME = 0.25f, BUMP_MIN_FORWARD_DIST = 64.5f, DEMO_RESPAWN_TIME = 3.f;
// RLConst::BTVehicle (RLConst.h:137-149)
constexpr float SUSPENSION_FORCE_SCALE_FRONT = 36.f - (1.f / 4.f), SUSPENSION_FORCE_SCALE_BACK = 54.f + (1.f / 4.f) + (1.5f / 100.f),
                SUSPENSION_STIFFNESS = 500.f, WHEELS_DAMPING_COMPRESSION = 25.f, WHEELS_DAMPING_RELAXATION = 40.f,
                MAX_SUSPENSION_TRAVEL = 12.f, SUSPENSION_SUBTRACTION = 0.05f;
constexpr float AIR_TORQUE_P = 130, AIR_TORQUE_Y = 95, AIR_TORQUE_R = 400;  // CAR_AIR_CONTROL_TORQUE (RLConst.h:188)
constexpr float AIR_DAMP_P = 30, AIR_DAMP_Y = 20, AIR_DAMP_R = 50;           // CAR_AIR_CONTROL_DAMPING (RLConst.h:190)
// boost pads (RLConst.h:192-253)
constexpr float PAD_CYL_HEIGHT = 95, PAD_CYL_RAD_BIG = 208, PAD_CYL_RAD_SMALL = 144, PAD_BOX_HEIGHT = 64, PAD_BOX_RAD_BIG = 160,
                PAD_BOX_RAD_SMALL = 120, PAD_COOLDOWN_BIG = 10, PAD_COOLDOWN_SMALL = 4, PAD_BOOST_BIG = 100, PAD_BOOST_SMALL = 12;
// Octane (CarConfig.cpp:20-70)
constexpr float HITBOX_X = 120.507f, HITBOX_Y = 86.6994f, HITBOX_Z = 38.6591f;
constexpr float HITBOX_OFF_X = 13.87566f, HITBOX_OFF_Y = 0.f, HITBOX_OFF_Z = 20.755f;
constexpr float WHEEL_RAD_FRONT = 12.50f, WHEEL_RAD_BACK = 15.00f;
constexpr float SUS_REST_FRONT = 38.755f, SUS_REST_BACK = 37.055f;
constexpr float WHEEL_FX = 51.25f, WHEEL_FY = 25.90f, WHEEL_FZ = 20.755f;
constexpr float WHEEL_BX = -33.75f, WHEEL_BY = 29.50f, WHEEL_BZ = 20.755f;
constexpr float DODGE_DEADZONE = 0.5f;  // CarConfig.h dodgeDeadzone
// Bullet solver settings (Arena.cpp:483-488; btContactSolverInfo.h:78-113)
constexpr int   SOLVER_ITERS = 10;
constexpr float ERP = 0.2f, ERP2 = 0.8f, SPLIT_TURN_ERP = 0.1f, RESTITUTION_VEL_THRESHOLD = 0.2f;
constexpr float WORLD_RESTITUTION = 0.3f, WORLD_FRICTION = 0.6f;  // Arena.cpp:502-508
}  // namespace K

// ---- piecewise-linear curves (RocketSim/src/Math/Math.cpp:5-38; tables RLConst.h:342-437) ------------
template <int N>
RLG_HD float curve(const float (&xs)[N], const float (&ys)[N], float in) {
    if (in <= xs[0]) return ys[0];
#pragma unroll
    for (int i = 1; i < N; i++) {
        if (xs[i] > in) {
            float range = xs[i] - xs[i - 1];
            float dv = ys[i] - ys[i - 1];
            float f = (in - xs[i - 1]) / range;
            return ys[i - 1] + dv * f;
        }
    }
    return ys[N - 1];
}
RLG_HD float curve_steer_angle(float v) {
    const float xs[6] = {0, 500, 1000, 1500, 1750, 3000};
    const float ys[6] = {0.53356f, 0.31930f, 0.18203f, 0.10570f, 0.08507f, 0.03454f};
    return curve(xs, ys, v);
}
RLG_HD float curve_powerslide_steer(float v) {
    const float xs[2] = {0, 2500}; const float ys[2] = {0.39235f, 0.12610f};
    return curve(xs, ys, v);
}
RLG_HD float curve_drive_torque(float v) {
    const float xs[3] = {0, 1400, 1410}; const float ys[3] = {1.0f, 0.1f, 0.0f};
    return curve(xs, ys, v);
}
RLG_HD float curve_non_sticky(float v) {
    const float xs[3] = {0, 0.7075f, 1}; const float ys[3] = {0.1f, 0.5f, 1.0f};
    return curve(xs, ys, v);
}
RLG_HD float curve_lat_friction(float v) {
    const float xs[2] = {0, 1}; const float ys[2] = {1.0f, 0.2f};
    return curve(xs, ys, v);
}
RLG_HD float curve_handbrake_long(float v) {
    const float xs[2] = {0, 1}; const float ys[2] = {0.5f, 0.9f};
    return curve(xs, ys, v);
}
RLG_HD float curve_ball_car_extra(float v) {
    const float xs[4] = {0, 500.f, 2300.f, 4600.f}; const float ys[4] = {0.65f, 0.65f, 0.55f, 0.30f};
    return curve(xs, ys, v);
}
RLG_HD float curve_bump_ground(float v) {
    const float xs[3] = {0.f, 1400.f, 2200.f}; const float ys[3] = {(5.f / 6.f), 1100.f, 1530.f};
    return curve(xs, ys, v);
}
RLG_HD float curve_bump_air(float v) {
    const float xs[3] = {0.f, 1400.f, 2200.f}; const float ys[3] = {(5.f / 6.f), 1390.f, 1945.f};
    return curve(xs, ys, v);
}
RLG_HD float curve_bump_up(float v) {
    const float xs[3] = {0.f, 1400.f, 2200.f}; const float ys[3] = {(2.f / 6.f), 278.f, 417.f};
    return curve(xs, ys, v);
}

// ---- flags (same bit values as include/rlgpu_state.h) ---------------------------------------------
enum : uint32_t {
    CF_ON_GROUND = 1u << 0, CF_WHEEL0 = 1u << 1, CF_HAS_JUMPED = 1u << 5, CF_HAS_DOUBLE_JUMPED = 1u << 6,
    CF_HAS_FLIPPED = 1u << 7, CF_IS_FLIPPING = 1u << 8, CF_IS_JUMPING = 1u << 9, CF_IS_SUPERSONIC = 1u << 10,
    CF_IS_AUTOFLIPPING = 1u << 11, CF_WORLD_CONTACT = 1u << 12, CF_IS_DEMOED = 1u << 13, CF_BALLHIT_VALID = 1u << 14,
    CF_ABSENT = 1u << 15          // an orange slot of a one-team env (spawnOpponents = false): no car here, ever
};

struct Controls {
    float throttle, steer, pitch, yaw, roll;
    bool jump, boost, handbrake;
};

// a rigid body in BT units (the slice of btRigidBody the tick touches)
struct Body {
    V3 pos;
    M3 rot;          // basis; columns forward/right/up
    V3 vel, angvel;
    V3 force, torque;  // m_totalForce / m_totalTorque, cleared each tick
    M3 inv_inertia_w;  // m_invInertiaTensorWorld (refreshed when the transform is set)
};

// per-wheel values that live only inside one tick
struct WheelTmp {
    V3 hard_point, contact_point, contact_normal;
    V3 impulse;
    float susp_len, susp_rel_vel, clipped_inv;
    int ground;  // -1 none, 0 static world, 1 ball, 2+k car k
    bool in_contact;
    // what the wheel's suspension force and friction impulse add to the car's velocities (btRigidBody::applyImpulse, btRigidBody.h:342-352): computed by
    // the wheel's own lane at the end of phase 1c -- they depend on the wheel and on the car's pose only --, added up by the car's lane in wheel order
    // (the sums are formed in the reference's order; only the products moved).  *_nz: the reference applies the impulse at all (it skips zero ones)
    bool susp_nz, fric_nz;
    V3 susp_dv, susp_dw, fric_dv, fric_dw;
};

// CarHot = what the car's control phase (arena_car.h:car_pre_tick_finish, every tick, on a register copy of the car) reads or writes; the rest of
// a Car is never touched there and stays out of that copy -- a whole-struct copy moved those 26 words LDS -> scratch -> LDS every tick, because a
// slice of a local that is only ever copied is not promoted to registers.
struct CarHot {
    Body b;
    uint32_t flags;
    V3 flip_rel_torque;
    float jump_time, flip_time, air_time, air_time_since_jump;
    float boost, time_spent_boosting, handbrake_val;
    float auto_flip_timer, auto_flip_torque_scale;
    V3 world_contact_normal;
    Controls ctl;
    float extra_pushback[4];
    float steer_angle, engine_force, brake;
    float lat_friction[4], long_friction[4];
};
struct Car : CarHot {
    float supersonic_time;
    int car_contact_other;  // car id (slot+1), 0 none
    float car_contact_cooldown, demo_respawn_timer;
    V3 bh_rel_pos, bh_ball_pos, bh_extra_hit_vel;  // uu
    int64_t bh_tick_hit, bh_tick_extra;
    Controls last;
    V3 vel_impulse_cache;  // BT
    bool frozen;  // transient: body disabled for the current tick (demoed at tick start)
};

struct Ball {
    Body b;
    V3 vel_impulse_cache;  // BT
};

struct Pad {   // 8 B: an env's 34 pads are LDS-resident on the device
    float cooldown;
    bool is_active;
    int8_t prev_locked;  // car id, 0 none
    int8_t cur_locked;   // car id during the tick
};

// ---- static world -----------------------------------------------------------------------------------
// Threaded BVH: besides its children every node knows its "escape" = the node a right-child-first depth-first walk visits
// after this node's subtree.  A walk is then `i = hit ? (leaf ? escape : right child) : escape` with no stack -- on the
// GPU a per-lane stack is either scratch memory or (what the compiler picked) a 32-way select chain per push/pop.
struct alignas(16) BvhNode {  // 32 B
    float minx, miny, minz;
    int32_t left_or_first;
    float maxx, maxy, maxz;
    uint32_t count_escape;  // bits 0-7: 0 => inner node with children left, left+1 ; >0 => leaf with that many triangles from `left_or_first`
                            // bits 8-31: escape node index, BVH_END when the walk is over
};
constexpr uint32_t BVH_END = 0xFFFFFFu;
RLG_HD int node_count(const BvhNode& n) { return (int)(n.count_escape & 0xffu); }
RLG_HD uint32_t node_escape(const BvhNode& n) { return n.count_escape >> 8; }
struct alignas(16) MeshTri {  // 64 B, BT units
    float v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z;
    // btTriangleInfo of this triangle (BulletCollision/CollisionShapes/btTriangleInfoMap.h:27-52, filled the way btGenerateInternalEdgeInfo
    // does, arena_mesh.cpp): dihedral angle over edge v0v1 / v1v2 / v2v0 towards the neighbour sharing it (2 pi = no neighbour) ...
    float edge_angle[3];
    uint32_t edge_flags;  // ... and TRI_INFO_V0V1_CONVEX = 1, V1V2_CONVEX = 2, V2V0_CONVEX = 4, V0V1_SWAP_NORMALB = 8, V1V2_SWAP = 16, V2V0_SWAP = 32;
                          // bit 31: the triangle has an info record at all (some edge is shared)
    uint32_t obj;         // the mesh OBJECT (.cmf file) the triangle belongs to: the reference makes one collision object, and with it one
                          // contact manifold per dynamic body, per file (Arena.cpp:1028-1054)
    uint32_t _pad1, _pad2;
};
// coarse occupancy grid over the arena volume: bit set <=> some mesh triangle overlaps the 256 uu cell.  Most queries
// (mid-field wheel rays, ball / hitbox AABBs) touch only empty cells and skip the BVH walk altogether.
constexpr int GRID_X = 34, GRID_Y = 48, GRID_Z = 10;
constexpr float GRID_CELL = 5.12f;                                  // BT (256 uu)
constexpr float GRID_MIN_X = -87.04f, GRID_MIN_Y = -122.88f, GRID_MIN_Z = -5.12f;
constexpr int GRID_WORDS = (GRID_X * GRID_Y * GRID_Z + 31) / 32;

// boost pad lookup words (arena_step.h:pad_table_fill): per pad (x + 8192) | (y + 8192) << 16 in uu, then per cell of the reference's
// 8 x 10 pad grid (BoostPadGrid.cpp:27-41) the pads of its 3 x 3 neighbourhood that a car in the cell can reach at all: up to three
// bytes (pad + 1, 0 = none) per word
constexpr int PAD_TAB_WORDS = 34 + 80;

struct MeshView {
    const BvhNode* nodes;      // global
    const MeshTri* tris;       // global
    const BvhNode* nodes_fast; // LDS-staged copy of the first n_fast nodes (device) or nullptr
    const uint32_t* grid;      // GRID_WORDS occupancy words (LDS on the device), or nullptr = always traverse
    const uint32_t* bp;        // global: the number of mesh objects, their boxes (6 floats each), then per cell of the reference's broadphase the
                               // mask of the objects listed there, see arena_mesh.cpp / arena_contact.h; nullptr = one object, listed everywhere
    int n_nodes, n_tris, n_fast;
};

// ---- MutatorConfig (RocketSim MutatorConfig.h:18-75), the part the stepper takes at run time: the scalars that change no collision shape, no mass and no
// mass (gravity, the boost / jump / ball-speed numbers, the timers, the demolition rules, the goal line, the world friction / restitution values).  One copy per env, part of its state
// (RlgpuArenaState::mutators): read where the reference reads _mutatorConfig, so no register is held for it between uses.  Car / ball mass and the ball's
// radius stay compiled in (the facade refuses non-default values for those: include/RLGymSim_CPP/RocketSim/Arena.h).
struct Mutators {
    float gravity_z;                 // uu/s^2
    float boost_accel_ground, boost_accel_air, boost_used_per_second;
    float jump_accel, jump_immediate_force;
    float ball_max_speed;            // uu/s
    float ball_damp_per_tick;        // powf(1 - ballDrag, 1 / 120) as the C library rounds it (btRigidBody::applyDamping); derived on the host
    float respawn_delay, bump_cooldown, pad_cooldown_big, pad_cooldown_small;
    float spawn_boost;               // carSpawnBoostAmount
    float ball_hit_extra_scale, bump_force_scale;
    float goal_threshold_y;          // goalBaseThresholdY
    float gravity_x, gravity_y;      // uu/s^2 (a sideways gravity: the same two body forces, Arena.cpp:25)
    float car_world_friction, car_world_restitution;     // the car-world manifold points' combined values (Arena.cpp:425-426)
    float ball_world_friction, ball_world_restitution;   // the ball body's own values: against the static world they combine as min(f, 0.6) / max(r, 0.3) (btManifoldResult.cpp:56-78, Arena.cpp:505-506)
    uint32_t flags;                  // MUT_*
};
constexpr uint32_t MUT_UNLIMITED_FLIPS = 1u, MUT_UNLIMITED_DOUBLE_JUMPS = 2u, MUT_DEMO_ON_CONTACT = 4u, MUT_DEMO_DISABLED = 8u, MUT_TEAM_DEMOS = 16u;
constexpr uint32_t MUT_RAY_PROXY_LISTS = 32u;   // not a MutatorConfig field: wheel rays are cast against every dynamic proxy the reference's broadphase lists for them (arena_world.h ray_ball_and_cars)
constexpr int MUTATOR_WORDS = 23;
RLG_HD Mutators mutators_default() {
    Mutators m;
    m.gravity_z = K::GRAVITY_Z; m.boost_accel_ground = K::BOOST_ACCEL_GROUND; m.boost_accel_air = K::BOOST_ACCEL_AIR; m.boost_used_per_second = K::BOOST_USED_PER_SECOND;
    m.jump_accel = K::JUMP_ACCEL; m.jump_immediate_force = K::JUMP_IMMEDIATE_FORCE; m.ball_max_speed = K::BALL_MAX_SPEED; m.ball_damp_per_tick = K::BALL_DAMP_PER_TICK;
    m.respawn_delay = K::DEMO_RESPAWN_TIME; m.bump_cooldown = K::BUMP_COOLDOWN_TIME; m.pad_cooldown_big = K::PAD_COOLDOWN_BIG; m.pad_cooldown_small = K::PAD_COOLDOWN_SMALL;
    m.spawn_boost = K::BOOST_SPAWN_AMOUNT; m.ball_hit_extra_scale = 1.f; m.bump_force_scale = 1.f; m.goal_threshold_y = K::GOAL_THRESHOLD_Y; m.flags = 0u;
    m.gravity_x = 0.f; m.gravity_y = 0.f; m.car_world_friction = K::CARWORLD_FRICTION; m.car_world_restitution = K::CARWORLD_RESTITUTION;
    m.ball_world_friction = K::BALL_FRICTION; m.ball_world_restitution = K::BALL_RESTITUTION;
    return m;
}

// events a tick can raise towards the gym layer (Gym.cpp:6-38 callbacks)
struct TickEvents {
    uint32_t bump_mask;  // bit i: car i bumped an opponent this tick ; bit 8+i : ... and it was a demo
};

template <int NC>
struct Arena {
    Ball ball;
    Car cars[NC];
    Pad pads[34];
    int64_t tick_count;
    int64_t ball_update_counter;
    uint32_t car_order;   // RlgpuArenaState::car_order: 4 bits per rank (slot + 1), 0 = slot order
    uint32_t ref_engine;  // RlgpuArenaHidden::ref_engine: 0 = the env draws from its own Philox streams; else the state of the reference thread's
                          // std::default_random_engine, from which the env then draws what the reference draws, as the reference draws it (parity tests)
    // btRSBroadphase's memory of its dynamic proxies (ball, cars), hidden state of the reference that no CarState shows: the cell of each
    // proxy's last setAabb (13 bits) and its arrival rank among the dynamic proxies (3 bits) -- arena_step.h bp_history_track.  All zero
    // = a fresh arena (what a state uploaded from the host starts as).
    uint16_t bp_hist[NC + 1];
    Mutators mut;         // MutatorConfig's run-time scalars (above); every env of a batch starts with the batch's (GymConfig::mutators)
};
// the car the arena's per-car loops visit k-th (Arena.cpp:716-812 iterates an unordered set of car pointers)
template <int NC>
RLG_HD int car_at_rank(const Arena<NC>& A, int k) { return A.car_order ? (int)((A.car_order >> (4 * k)) & 15u) - 1 : k; }
RLG_HD uint32_t car_order_checked(uint32_t order, int nc) {   // a permutation of the slots, or 0
    uint32_t seen = 0u;
    for (int k = 0; k < nc; k++) { const int s = (int)((order >> (4 * k)) & 15u) - 1; if (s < 0 || s >= nc) return 0u; seen |= 1u << s; }
    return seen == (1u << nc) - 1u && (order >> (4 * nc)) == 0u ? order : 0u;
}

}  // namespace rlg
