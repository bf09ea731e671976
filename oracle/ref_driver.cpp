// oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// A thin C-ABI shim over the REAL reference simulator (RocketSim + RLGymSim_CPP compiled from the
// sources where they lie under /root/reference by oracle/Makefile -> oracle/_ref/libref_oracle.so).
// It lets the tests (a) generate the golden fixtures under tests/golden/, (b) compare the HIP path with
// the reference itself on the GPU box (the prebuilt .so travels with gpurun), and (c) time the
// reference's CPU path for bench.py's cpu_baseline (kind "reference").
//
// All state crosses this boundary as RlgpuArenaState (include/rlgpu_state.h) in SLOT order
// (slot 2k = blue k, slot 2k+1 = orange k, car id = slot+1); the reference's own
// std::unordered_set<Car*> iteration order (Arena.h:35) is hidden in here.
#include <RLGymSim_CPP/Gym.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CombinedReward.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/ZeroSumReward.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBSPadded.h>
#include <RLGymSim_CPP/Utils/StateSetters/RandomState.h>
#include <RLGymSim_CPP/Utils/StateSetters/KickoffState.h>
#include <RLGymPPO_CPP/Threading/GameInst.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>

#include <bullet3-3.24/BulletCollision/CollisionShapes/btTriangleShape.h>
#include <bullet3-3.24/BulletCollision/CollisionShapes/btBoxShape.h>
#include <bullet3-3.24/BulletCollision/NarrowPhaseCollision/btGjkPairDetector.h>
#include <bullet3-3.24/BulletCollision/NarrowPhaseCollision/btVoronoiSimplexSolver.h>
#include <bullet3-3.24/BulletCollision/NarrowPhaseCollision/btGjkEpaPenetrationDepthSolver.h>
#include <bullet3-3.24/BulletCollision/BroadphaseCollision/btRSBroadphase.h>
#include <bullet3-3.24/BulletCollision/CollisionDispatch/btSimulationIslandManager.h>
#include <bullet3-3.24/BulletCollision/CollisionDispatch/btInternalEdgeUtility.h>
#include <bullet3-3.24/BulletCollision/CollisionDispatch/btCollisionObjectWrapper.h>
#include <bullet3-3.24/BulletCollision/CollisionDispatch/btCollisionWorld.h>
#include <bullet3-3.24/BulletCollision/CollisionShapes/btSphereShape.h>
#include <bullet3-3.24/BulletCollision/CollisionDispatch/btBoxBoxDetector.h>
#include "../include/rlgpu_state.h"

#include <cstring>
#include <thread>
#include <atomic>
#include <chrono>

using namespace RLGSC;

namespace {

// Exchange-layout slots: blue cars on the even ones, orange on the odd ones, in AddCar order (car id = slot + 1 for Gym's blue / orange
// alternation).  An arena WITHOUT orange cars -- Match(..., spawnOpponents = false) -- has ids 1..P for its P blue cars: slot 2 (id - 1),
// and its odd slots are empty (flagged RLGPU_CF_ABSENT in the exchange state).
bool OneTeam(Arena* a) { for (Car* c : a->_cars) if (c->team != Team::BLUE) return false; return !a->_cars.empty(); }
int SlotOfCar(Arena* a, uint32_t id) { return OneTeam(a) ? 2 * ((int)id - 1) : (int)id - 1; }
int SlotCount(Arena* a) { return OneTeam(a) ? 2 * (int)a->_cars.size() : (int)a->_cars.size(); }
Car* CarBySlot(Arena* a, int slot) {
    if (OneTeam(a)) { if (slot % 2) return nullptr; slot /= 2; }
    auto it = a->_carIDMap.find((uint32_t)slot + 1);
    return it == a->_carIDMap.end() ? nullptr : it->second;
}

void V3(float* o, const Vec& v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
Vec  toV(const float* p) { return Vec(p[0], p[1], p[2]); }

void CtrlToArr(const CarControls& c, float* o) {
    o[0] = c.throttle; o[1] = c.steer; o[2] = c.pitch; o[3] = c.yaw; o[4] = c.roll;
    o[5] = c.jump; o[6] = c.boost; o[7] = c.handbrake;
}
CarControls ArrToCtrl(const float* p) {
    CarControls c; c.throttle = p[0]; c.steer = p[1]; c.pitch = p[2]; c.yaw = p[3]; c.roll = p[4];
    c.jump = p[5] != 0; c.boost = p[6] != 0; c.handbrake = p[7] != 0; return c;
}

void GetArenaPhys(Arena* a, RlgpuArenaState* s) {
    memset(s, 0, sizeof(*s));
    s->num_cars = SlotCount(a);
    {   // the order in which Arena::Step's `for (Car* car : _cars)` loops visit the cars (an unordered_set of pointers: Arena.h:35), as slots;
        // absent slots of a one-team arena go last so that the word stays a permutation
        uint32_t order = 0; int rank = 0; uint32_t seen = 0;
        for (Car* car : a->_cars) { const int slot = SlotOfCar(a, car->id); order |= (uint32_t)(slot + 1) << (4 * rank++); seen |= 1u << slot; }
        for (int slot = 0; slot < s->num_cars; slot++) if (!((seen >> slot) & 1u)) order |= (uint32_t)(slot + 1) << (4 * rank++);
        s->car_order = order;
    }
    s->tick_count = (int64_t)a->tickCount;
    BallState bs = a->ball->GetState();
    s->ball_update_counter = (int64_t)bs.updateCounter;
    V3(s->ball.pos, bs.pos); V3(s->ball.vel, bs.vel); V3(s->ball.ang_vel, bs.angVel);
    V3(s->ball.vel_impulse_cache, a->ball->_velocityImpulseCache * BT_TO_UU);
    // BallState::rotMat (Ball.cpp:27-30).  Nothing else of RlgpuArenaHidden is read out here: valid = 0
    V3(s->hidden.ball_rot, bs.rotMat.forward); V3(s->hidden.ball_rot + 3, bs.rotMat.right); V3(s->hidden.ball_rot + 6, bs.rotMat.up);
    s->hidden.valid = 0;
    {   // the arena's MutatorConfig, as far as RlgpuMutators carries it (the per-tick damping factor as btRigidBody::applyDamping forms it)
        const MutatorConfig& mc = a->GetMutatorConfig();
        RlgpuMutators& m = s->mutators;
        m.gravity_z = mc.gravity.z; m.boost_accel_ground = mc.boostAccelGround; m.boost_accel_air = mc.boostAccelAir; m.boost_used_per_second = mc.boostUsedPerSecond;
        m.jump_accel = mc.jumpAccel; m.jump_immediate_force = mc.jumpImmediateForce; m.ball_max_speed = mc.ballMaxSpeed;
        m.ball_damp_per_tick = powf(1.0f - mc.ballDrag, 1.0f / 120.0f);
        m.respawn_delay = mc.respawnDelay; m.bump_cooldown_time = mc.bumpCooldownTime; m.boost_pad_cooldown_big = mc.boostPadCooldown_Big; m.boost_pad_cooldown_small = mc.boostPadCooldown_Small;
        m.car_spawn_boost_amount = mc.carSpawnBoostAmount; m.ball_hit_extra_force_scale = mc.ballHitExtraForceScale; m.bump_force_scale = mc.bumpForceScale;
        m.goal_base_threshold_y = mc.goalBaseThresholdY;
        m.gravity_x = mc.gravity.x; m.gravity_y = mc.gravity.y; m.car_world_friction = mc.carWorldFriction; m.car_world_restitution = mc.carWorldRestitution;
        m.ball_world_friction = mc.ballWorldFriction; m.ball_world_restitution = mc.ballWorldRestitution;
        m.flags = (mc.unlimitedFlips ? RLGPU_MUT_UNLIMITED_FLIPS : 0u) | (mc.unlimitedDoubleJumps ? RLGPU_MUT_UNLIMITED_DOUBLE_JUMPS : 0u) |
                  (mc.demoMode == DemoMode::ON_CONTACT ? RLGPU_MUT_DEMO_ON_CONTACT : mc.demoMode == DemoMode::DISABLED ? RLGPU_MUT_DEMO_DISABLED : 0u) |
                  (mc.enableTeamDemos ? RLGPU_MUT_TEAM_DEMOS : 0u);
        s->hidden.valid |= RLGPU_HIDDEN_MUTATORS;
    }
    for (int i = 0; i < s->num_cars; i++) {
        Car* car = CarBySlot(a, i);
        RlgpuCarState& o = s->cars[i];
        if (!car) {   // empty slot of a one-team arena, as the device parks it (csrc/arena_gym.h park_absent_players)
            o.flags = RLGPU_CF_IS_DEMOED | RLGPU_CF_ABSENT; o.demo_respawn_timer = 1e30f; o.pos[2] = -10000.f;
            o.rot[0] = o.rot[4] = o.rot[8] = 1.f; o.bh_tick_hit = -1; o.bh_tick_extra = -1;
            continue;
        }
        CarState cs = car->GetState();
        V3(o.pos, cs.pos); V3(o.rot, cs.rotMat.forward); V3(o.rot + 3, cs.rotMat.right); V3(o.rot + 6, cs.rotMat.up);
        V3(o.vel, cs.vel); V3(o.ang_vel, cs.angVel);
        uint32_t f = 0;
        if (cs.isOnGround) f |= RLGPU_CF_ON_GROUND;
        for (int w = 0; w < 4; w++) if (cs.wheelsWithContact[w]) f |= (RLGPU_CF_WHEEL0 << w);
        if (cs.hasJumped) f |= RLGPU_CF_HAS_JUMPED;
        if (cs.hasDoubleJumped) f |= RLGPU_CF_HAS_DOUBLE_JUMPED;
        if (cs.hasFlipped) f |= RLGPU_CF_HAS_FLIPPED;
        if (cs.isFlipping) f |= RLGPU_CF_IS_FLIPPING;
        if (cs.isJumping) f |= RLGPU_CF_IS_JUMPING;
        if (cs.isSupersonic) f |= RLGPU_CF_IS_SUPERSONIC;
        if (cs.isAutoFlipping) f |= RLGPU_CF_IS_AUTOFLIPPING;
        if (cs.worldContact.hasContact) f |= RLGPU_CF_WORLD_CONTACT;
        if (cs.isDemoed) f |= RLGPU_CF_IS_DEMOED;
        if (cs.ballHitInfo.isValid) f |= RLGPU_CF_BALLHIT_VALID;
        o.flags = f;
        V3(o.flip_rel_torque, cs.flipRelTorque);
        o.jump_time = cs.jumpTime; o.flip_time = cs.flipTime;
        o.air_time = cs.airTime; o.air_time_since_jump = cs.airTimeSinceJump;
        o.boost = cs.boost; o.time_spent_boosting = cs.timeSpentBoosting;
        o.supersonic_time = cs.supersonicTime; o.handbrake_val = cs.handbrakeVal;
        o.auto_flip_timer = cs.autoFlipTimer; o.auto_flip_torque_scale = cs.autoFlipTorqueScale;
        V3(o.world_contact_normal, cs.worldContact.contactNormal);
        o.car_contact_other_id = (int32_t)cs.carContact.otherCarID;
        o.car_contact_cooldown = cs.carContact.cooldownTimer;
        o.demo_respawn_timer = cs.demoRespawnTimer;
        V3(o.bh_rel_pos, cs.ballHitInfo.relativePosOnBall); V3(o.bh_ball_pos, cs.ballHitInfo.ballPos);
        V3(o.bh_extra_hit_vel, cs.ballHitInfo.extraHitVel);
        o.bh_tick_hit = (int64_t)cs.ballHitInfo.tickCountWhenHit;
        o.bh_tick_extra = (int64_t)cs.ballHitInfo.tickCountWhenExtraImpulseApplied;
        CtrlToArr(cs.lastControls, o.last_controls);
        CtrlToArr(car->controls, o.controls);
        V3(o.vel_impulse_cache, car->_velocityImpulseCache * BT_TO_UU);
        for (int w = 0; w < 4; w++) {
            const auto& wi = car->_bulletVehicle.m_wheelInfo[w];
            o.extra_pushback[w] = wi.m_extraPushback;
            o.wheel_lat_friction[w] = wi.m_latFriction; o.wheel_long_friction[w] = wi.m_longFriction;
        }
        o.wheel_steer_angle = car->_bulletVehicle.m_wheelInfo[0].m_steerAngle;
        o.wheel_engine_force = car->_bulletVehicle.m_wheelInfo[0].m_engineForce;
        o.wheel_brake = car->_bulletVehicle.m_wheelInfo[0].m_brake;
    }
    for (int i = 0; i < RLGPU_NUM_PADS && i < (int)a->_boostPads.size(); i++) {
        BoostPadState ps = a->_boostPads[i]->GetState();
        s->pads[i].cooldown = ps.cooldown;
        s->pads[i].is_active = ps.isActive;
        s->pads[i].prev_locked_car_id = (int32_t)ps.prevLockedCarID;
    }
}

void SetArenaPhys(Arena* a, const RlgpuArenaState* s, bool setPads) {
    a->tickCount = (uint64_t)s->tick_count;
    BallState bs = {};
    bs.pos = toV(s->ball.pos); bs.vel = toV(s->ball.vel); bs.angVel = toV(s->ball.ang_vel);
    {   // (an all-zero basis = a caller that knows nothing of the appended block: a default BallState)
        bool all_zero = true;
        for (int q = 0; q < 9; q++) all_zero = all_zero && s->hidden.ball_rot[q] == 0.f;
        if (!all_zero) { bs.rotMat.forward = toV(s->hidden.ball_rot); bs.rotMat.right = toV(s->hidden.ball_rot + 3); bs.rotMat.up = toV(s->hidden.ball_rot + 6); }
    }
    a->ball->SetState(bs);
    a->ball->_velocityImpulseCache = toV(s->ball.vel_impulse_cache) * UU_TO_BT;
    a->ball->_internalState.updateCounter = (uint64_t)s->ball_update_counter;
    for (int i = 0; i < s->num_cars; i++) {
        Car* car = CarBySlot(a, i);
        if (!car) continue;
        const RlgpuCarState& o = s->cars[i];
        CarState cs = {};
        cs.pos = toV(o.pos); cs.rotMat.forward = toV(o.rot); cs.rotMat.right = toV(o.rot + 3); cs.rotMat.up = toV(o.rot + 6);
        cs.vel = toV(o.vel); cs.angVel = toV(o.ang_vel);
        uint32_t f = o.flags;
        cs.isOnGround = f & RLGPU_CF_ON_GROUND;
        for (int w = 0; w < 4; w++) cs.wheelsWithContact[w] = f & (RLGPU_CF_WHEEL0 << w);
        cs.hasJumped = f & RLGPU_CF_HAS_JUMPED; cs.hasDoubleJumped = f & RLGPU_CF_HAS_DOUBLE_JUMPED;
        cs.hasFlipped = f & RLGPU_CF_HAS_FLIPPED; cs.isFlipping = f & RLGPU_CF_IS_FLIPPING;
        cs.isJumping = f & RLGPU_CF_IS_JUMPING; cs.isSupersonic = f & RLGPU_CF_IS_SUPERSONIC;
        cs.isAutoFlipping = f & RLGPU_CF_IS_AUTOFLIPPING; cs.worldContact.hasContact = f & RLGPU_CF_WORLD_CONTACT;
        cs.isDemoed = f & RLGPU_CF_IS_DEMOED; cs.ballHitInfo.isValid = f & RLGPU_CF_BALLHIT_VALID;
        cs.flipRelTorque = toV(o.flip_rel_torque);
        cs.jumpTime = o.jump_time; cs.flipTime = o.flip_time; cs.airTime = o.air_time; cs.airTimeSinceJump = o.air_time_since_jump;
        cs.boost = o.boost; cs.timeSpentBoosting = o.time_spent_boosting; cs.supersonicTime = o.supersonic_time;
        cs.handbrakeVal = o.handbrake_val; cs.autoFlipTimer = o.auto_flip_timer; cs.autoFlipTorqueScale = o.auto_flip_torque_scale;
        cs.worldContact.contactNormal = toV(o.world_contact_normal);
        cs.carContact.otherCarID = (uint32_t)o.car_contact_other_id; cs.carContact.cooldownTimer = o.car_contact_cooldown;
        cs.demoRespawnTimer = o.demo_respawn_timer;
        cs.ballHitInfo.relativePosOnBall = toV(o.bh_rel_pos); cs.ballHitInfo.ballPos = toV(o.bh_ball_pos);
        cs.ballHitInfo.extraHitVel = toV(o.bh_extra_hit_vel);
        cs.ballHitInfo.tickCountWhenHit = (uint64_t)o.bh_tick_hit;
        cs.ballHitInfo.tickCountWhenExtraImpulseApplied = (uint64_t)o.bh_tick_extra;
        cs.lastControls = ArrToCtrl(o.last_controls);
        car->SetState(cs);
        car->controls = ArrToCtrl(o.controls);
        car->_velocityImpulseCache = toV(o.vel_impulse_cache) * UU_TO_BT;
        for (int w = 0; w < 4; w++) {
            auto& wi = car->_bulletVehicle.m_wheelInfo[w];
            wi.m_extraPushback = o.extra_pushback[w];
            wi.m_latFriction = o.wheel_lat_friction[w]; wi.m_longFriction = o.wheel_long_friction[w];
            wi.m_engineForce = o.wheel_engine_force; wi.m_brake = o.wheel_brake;
            wi.m_steerAngle = (w < 2) ? o.wheel_steer_angle : 0.f;
        }
    }
    if (setPads) {
        for (int i = 0; i < RLGPU_NUM_PADS && i < (int)a->_boostPads.size(); i++) {
            BoostPadState ps = {};
            ps.cooldown = s->pads[i].cooldown; ps.isActive = s->pads[i].is_active;
            ps.prevLockedCarID = (uint32_t)s->pads[i].prev_locked_car_id;
            a->_boostPads[i]->SetState(ps);
        }
    }
}

// A state setter that installs a caller-provided state (the reference has no such built-in; the
// plugin surface allows it: StateSetter.h:5-10).
class FixedStateSetter : public StateSetter {
public:
    RlgpuArenaState next = {};
    bool kickoffFirst = false;
    virtual GameState ResetState(Arena* arena) {
        SetArenaPhys(arena, &next, false);
        return GameState(arena);
    }
};

struct RefGym {
    Match* match = nullptr;
    Gym* gym = nullptr;
    FixedStateSetter* setter = nullptr;
    NoTouchCondition* noTouch = nullptr;
    EventReward* eventReward = nullptr;
    std::vector<RewardFunction*> ownedRewards;
    std::vector<TerminalCondition*> conds;
    OBSBuilder* obs = nullptr;
    ActionParser* parser = nullptr;
    RewardFunction* rootReward = nullptr;
    int nPlayers = 0;
};

void FillGymState(RefGym* g, RlgpuArenaState* s) {
    const GameState& st = g->gym->prevState;
    RlgpuGymState& o = s->gym;
    o.score_line[0] = st.scoreLine[0]; o.score_line[1] = st.scoreLine[1];
    o.last_touch_car_id = st.lastTouchCarID > 0 ? SlotOfCar(g->gym->arena, (uint32_t)st.lastTouchCarID) + 1 : st.lastTouchCarID;
    o.last_tick_count = (int64_t)st.lastTickCount;
    o.no_touch_steps = g->noTouch ? g->noTouch->stepsSinceTouch : 0;
    const GameEventTracker& t = g->gym->eventTracker;
    o.shot_cooldown = t._shotCooldown; o.ball_shot = t._ballShot; o.ball_shot_goal_team = (uint8_t)t._ballShotGoalTeam;
    o.ball_scored_last = t._ballScoredLast; o.last_ball_update_count = (int64_t)t._lastBallUpdateCount;
    for (auto& p : st.players) {
        int slot = SlotOfCar(g->gym->arena, p.carId);
        RlgpuPlayerGymState& q = o.players[slot];
        q.match_goals = p.matchGoals; q.match_saves = p.matchSaves; q.match_assists = p.matchAssists;
        q.match_shots = p.matchShots; q.match_shot_passes = p.matchShotPasses; q.match_bumps = p.matchBumps;
        q.match_demos = p.matchDemos; q.boost_pickups = p.boostPickups;
        if (g->eventReward) {
            auto it = g->eventReward->lastRegisteredValues.find((int)p.carId);
            if (it != g->eventReward->lastRegisteredValues.end())
                for (int k = 0; k < RLGPU_NUM_EVENT_VALS; k++) q.event_last[k] = it->second.vals[k];
        }
    }
    for (size_t i = 0; i < st.players.size() && i < g->match->prevActions.size(); i++) {
        int slot = SlotOfCar(g->gym->arena, st.players[i].carId);
        for (int k = 0; k < 8; k++) o.players[slot].prev_action[k] = g->match->prevActions[i][k];
    }
}

}  // namespace

extern "C" {

// Initialise RocketSim with ONE synthetic soccar mesh given in uu (converted to the .cmf blob format:
// i32 nTris, i32 nVerts, tris, verts in Bullet units; CollisionMeshFile.cpp:11-36). Returns 0 on success.
int ref_init(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris) {
    if (RocketSim::GetStage() == RocketSim::RocketSimStage::INITIALIZED) return 0;
    std::vector<byte> blob(8 + (size_t)n_tris * 12 + (size_t)n_verts * 12);
    byte* p = blob.data();
    int32_t nt = n_tris, nv = n_verts;
    memcpy(p, &nt, 4); p += 4; memcpy(p, &nv, 4); p += 4;
    memcpy(p, tris, (size_t)n_tris * 12); p += (size_t)n_tris * 12;
    for (int i = 0; i < n_verts * 3; i++) { float v = verts_uu[i] * UU_TO_BT; memcpy(p, &v, 4); p += 4; }
    std::map<GameMode, std::vector<RocketSim::FileData>> m;
    m[GameMode::SOCCAR] = { blob };
    try { RocketSim::InitFromMem(m, true); } catch (std::exception& e) { fprintf(stderr, "ref_init: %s\n", e.what()); return -1; }
    return 0;
}

// RocketSim::Init on a directory of .cmf files (<dir>/soccar/*.cmf), the reference's own loader (RS/RocketSim.cpp:70-212): one
// btBvhTriangleMeshShape and one static rigid body per file (Arena.cpp:1028-1054).  Used for meshes of several files (the tessellated
// bench leg, the per-file parity fixtures); a process can be initialised once.
// (RocketSim::Init takes the files in std::filesystem::directory_iterator order, which is unspecified; the objects' creation order decides
// the order of a car's world manifolds, so this loader reads the same files in NAME order -- what rlgpu_env_load_cmf_dir does -- and hands
// them to RocketSim::InitFromMem, which is all Init does with them.)
int ref_init_dir(const char* dir) {
    if (RocketSim::GetStage() == RocketSim::RocketSimStage::INITIALIZED) return 0;
    std::vector<std::filesystem::path> files;
    const std::filesystem::path folder = std::filesystem::path(dir) / "soccar";
    if (!std::filesystem::exists(folder)) { fprintf(stderr, "ref_init_dir: no %s\n", folder.string().c_str()); return -1; }
    for (auto& entry : std::filesystem::directory_iterator(folder)) if (entry.path().extension() == ".cmf") files.push_back(entry.path());
    std::sort(files.begin(), files.end());
    std::map<GameMode, std::vector<RocketSim::FileData>> m;
    for (auto& f : files) { DataStreamIn in(f, false); m[GameMode::SOCCAR].push_back(in.data); }
    try { RocketSim::InitFromMem(m, true); } catch (std::exception& e) { fprintf(stderr, "ref_init_dir: %s\n", e.what()); return -1; }
    return 0;
}

// The order in which the arena mesh's triangles are handed to a convex body's collision callback when everything overlaps
// (btBvhTriangleMeshShape::processAllTriangles): out[i] = triangle index visited i-th.  Returns the triangle count.
int ref_mesh_visit_order(int32_t* out, int cap) {
    struct Rec : public btTriangleCallback {
        int32_t* out; int cap; int n = 0;
        void processTriangle(btVector3*, int, int triangleIndex) override { if (n < cap) out[n] = triangleIndex; n++; }
    } rec;
    rec.out = out; rec.cap = cap;
    auto& shapes = RocketSim::GetArenaCollisionShapes(GameMode::SOCCAR);
    if (shapes.empty()) return -1;
    const btVector3 lo(-1e9f, -1e9f, -1e9f), hi(1e9f, 1e9f, 1e9f);
    shapes[0]->processAllTriangles(&rec, lo, hi);
    return rec.n;
}

// One convex-convex query exactly as btConvexConvexAlgorithm::processCollision sets it up for a hitbox child against one mesh triangle
// (btConvexConvexAlgorithm.cpp:268-320,508: btGjkPairDetector + btVoronoiSimplexSolver + btGjkEpaPenetrationDepthSolver, maximum
// distance = both margins + the breaking threshold): a btBoxShape built from the full half extents (its constructor and setSafeMargin
// make the core and the margin RocketSim's hitbox has), transform (pos, rot row-major), against btTriangleShape(tri) with margin
// tri_margin in the identity frame.  out: normalOnBInWorld[3], pointInWorld[3], depth, box margin.  Returns 1 when the detector
// reported a point.  Unit-level oracle for csrc/arena_gjk.h (tests/test_oracle_golden.py, tools/gjk_fuzz.py).
int ref_gjk_box_triangle(const float* half3, const float* pos3, const float* rot9, const float* tri9, float tri_margin, float breaking, float* out8) {
    btBoxShape box(btVector3(half3[0], half3[1], half3[2]));
    btTriangleShape tri(btVector3(tri9[0], tri9[1], tri9[2]), btVector3(tri9[3], tri9[4], tri9[5]), btVector3(tri9[6], tri9[7], tri9[8]));
    tri.setMargin(tri_margin);
    btVoronoiSimplexSolver simplex;
    btGjkEpaPenetrationDepthSolver pd;
    btGjkPairDetector det(&box, &tri, &simplex, &pd);
    det.setMinkowskiA(&box); det.setMinkowskiB(&tri);
    btGjkPairDetector::ClosestPointInput input;
    input.m_maximumDistanceSquared = box.getMargin() + tri.getMargin() + breaking;
    input.m_maximumDistanceSquared *= input.m_maximumDistanceSquared;
    btMatrix3x3 basis(rot9[0], rot9[1], rot9[2], rot9[3], rot9[4], rot9[5], rot9[6], rot9[7], rot9[8]);
    input.m_transformA = btTransform(basis, btVector3(pos3[0], pos3[1], pos3[2]));
    input.m_transformB.setIdentity();
    struct Res : public btDiscreteCollisionDetectorInterface::Result {
        bool has = false; btVector3 n, p; btScalar d = 0;
        void setShapeIdentifiersA(int, int) override {}
        void setShapeIdentifiersB(int, int) override {}
        void addContactPoint(const btVector3& normalOnBInWorld, const btVector3& pointInWorld, btScalar depth) override { has = true; n = normalOnBInWorld; p = pointInWorld; d = depth; }
    } res;
    det.getClosestPoints(input, res);
    for (int k = 0; k < 3; k++) { out8[k] = res.has ? (float)res.n[k] : 0.f; out8[3 + k] = res.has ? (float)res.p[k] : 0.f; }
    out8[6] = res.has ? (float)res.d : 0.f; out8[7] = box.getMargin();
    return res.has ? 1 : 0;
}

// btBoxBoxDetector (ODE's dBoxBox2) on two boxes of the same full half extents, as btBoxBoxCollisionAlgorithm.cpp:55-70 runs it for two hitbox
// children: every point it reports, in order.  out: n x (normalOnBInWorld[3], pointInWorld[3], depth).  Unit-level oracle for
// csrc/arena_world.h:box_box_ode.
int ref_box_box(const float* half3, const float* pos1, const float* rot1, const float* pos2, const float* rot2, float* out, int cap) {
    btBoxShape b1(btVector3(half3[0], half3[1], half3[2])), b2(btVector3(half3[0], half3[1], half3[2]));
    btBoxBoxDetector det(&b1, &b2);
    btDiscreteCollisionDetectorInterface::ClosestPointInput input;
    input.m_maximumDistanceSquared = BT_LARGE_FLOAT;
    input.m_transformA = btTransform(btMatrix3x3(rot1[0], rot1[1], rot1[2], rot1[3], rot1[4], rot1[5], rot1[6], rot1[7], rot1[8]), btVector3(pos1[0], pos1[1], pos1[2]));
    input.m_transformB = btTransform(btMatrix3x3(rot2[0], rot2[1], rot2[2], rot2[3], rot2[4], rot2[5], rot2[6], rot2[7], rot2[8]), btVector3(pos2[0], pos2[1], pos2[2]));
    struct Res : public btDiscreteCollisionDetectorInterface::Result {
        float* out; int cap; int n = 0;
        void setShapeIdentifiersA(int, int) override {}
        void setShapeIdentifiersB(int, int) override {}
        void addContactPoint(const btVector3& nb, const btVector3& p, btScalar depth) override {
            if (n < cap) { float* o = out + 7 * n; o[0] = nb[0]; o[1] = nb[1]; o[2] = nb[2]; o[3] = p[0]; o[4] = p[1]; o[5] = p[2]; o[6] = depth; }
            n++;
        }
    } res; res.out = out; res.cap = cap;
    det.getClosestPoints(input, res);
    return res.n;
}

// btCollisionWorld::rayTestSingle on one convex object -- what a wheel's suspension ray meets when another car's hitbox child or the ball is
// in its way (btDefaultVehicleRaycaster.cpp:34-52 -> btCollisionWorld::rayTest -> rayTestSingleInternal: btSubsimplexConvexCast of a point
// against the shape, btCollisionWorld.cpp:267-310).  radius > 0: a btSphereShape, else a btBoxShape(half3).  out4 = hit fraction, m_hitNormalWorld.
// Unit-level oracle for csrc/arena_simplex.h:ray_convex_cast.
int ref_ray_convex(const float* from3, const float* to3, const float* half3, float radius, const float* pos3, const float* rot9, float* out4) {
    btBoxShape box(btVector3(half3[0], half3[1], half3[2]));
    btSphereShape sph(radius);
    btCollisionShape* shape = radius > 0.f ? (btCollisionShape*)&sph : (btCollisionShape*)&box;
    btCollisionObject obj; obj.setCollisionShape(shape);
    btMatrix3x3 basis(rot9[0], rot9[1], rot9[2], rot9[3], rot9[4], rot9[5], rot9[6], rot9[7], rot9[8]);
    btTransform tr(basis, btVector3(pos3[0], pos3[1], pos3[2]));
    obj.setWorldTransform(tr);
    const btVector3 from(from3[0], from3[1], from3[2]), to(to3[0], to3[1], to3[2]);
    btTransform tf, tt; tf.setIdentity(); tf.setOrigin(from); tt.setIdentity(); tt.setOrigin(to);
    btCollisionWorld::ClosestRayResultCallback cb(from, to, nullptr);   // (RocketSim adds the ignored object)
    btCollisionWorld::rayTestSingle(tf, tt, &obj, shape, tr, cb);
    out4[0] = cb.m_closestHitFraction; out4[1] = cb.m_hitNormalWorld[0]; out4[2] = cb.m_hitNormalWorld[1]; out4[3] = cb.m_hitNormalWorld[2];
    return cb.hasHit() ? 1 : 0;
}

// btAdjustInternalEdgeContacts as the contact-added callback runs it (Arena.cpp:275-279) on one new point against triangle `tri_index`
// of the arena mesh ref_init loaded (its btTriangleInfoMap comes from RocketSim.cpp:168-170): the point btManifoldResult::addContactPoint
// builds from (normalOnBInWorld n, pointInWorld pb, depth) with the mesh body at the identity (btManifoldResult.cpp:112-150).
// out7 = normalWorldOnB[3], positionWorldOnB[3], distance.  Unit-level oracle for csrc/arena_world.h:adjust_internal_edge.
int ref_adjust_internal_edge(int tri_index, const float* tri9, const float* pb3, const float* n3, float depth, float* out7) {
    auto& shapes = RocketSim::GetArenaCollisionShapes(GameMode::SOCCAR);
    if (shapes.empty()) return -1;
    btCollisionObject mesh_obj; mesh_obj.setCollisionShape(shapes[0]);
    btTransform id; id.setIdentity(); mesh_obj.setWorldTransform(id);
    btTriangleShape tm(btVector3(tri9[0], tri9[1], tri9[2]), btVector3(tri9[3], tri9[4], tri9[5]), btVector3(tri9[6], tri9[7], tri9[8]));
    btCollisionObjectWrapper tri_wrap(nullptr, &tm, &mesh_obj, mesh_obj.getWorldTransform(), 0, tri_index);
    btCollisionObject other; other.setWorldTransform(id);
    btCollisionObjectWrapper other_wrap(nullptr, nullptr, &other, other.getWorldTransform(), -1, -1);
    const btVector3 n(n3[0], n3[1], n3[2]), pb(pb3[0], pb3[1], pb3[2]);
    const btVector3 pa = pb + n * depth;
    btManifoldPoint cp(pa, pb, n, depth);   // local points = world points: both bodies at the identity
    cp.m_positionWorldOnA = pa; cp.m_positionWorldOnB = pb;
    btAdjustInternalEdgeContacts(cp, &tri_wrap, &other_wrap, 0, tri_index);
    for (int k = 0; k < 3; k++) { out7[k] = cp.m_normalWorldOnB[k]; out7[3 + k] = cp.m_positionWorldOnB[k]; }
    out7[6] = cp.m_distance1;
    return 0;
}

// the btTriangleInfo record btGenerateInternalEdgeInfo made for triangle `tri_index` of the arena mesh: out4 = flags, edge angles V0V1, V1V2, V2V0.  0 = no record.
int ref_triangle_info(int tri_index, float* out4) {
    auto& shapes = RocketSim::GetArenaCollisionShapes(GameMode::SOCCAR);
    if (shapes.empty()) return -1;
    btTriangleInfoMap* map = (btTriangleInfoMap*)shapes[0]->getTriangleInfoMap();
    btTriangleInfo* info = map ? map->find(tri_index)   /* btGetHash(partId 0, index) = index, btInternalEdgeUtility.cpp:32-36 */ : nullptr;
    if (!info) return 0;
    out4[0] = (float)info->m_flags; out4[1] = info->m_edgeV0V1Angle; out4[2] = info->m_edgeV1V2Angle; out4[3] = info->m_edgeV2V0Angle;
    return 1;
}

int ref_state_size() { return (int)sizeof(RlgpuArenaState); }

void* ref_arena_new(int team_size) {
    Arena* a = Arena::Create(GameMode::SOCCAR);
    for (int i = 0; i < team_size; i++) { a->AddCar(Team::BLUE); a->AddCar(Team::ORANGE); }
    return a;
}
// ... with every car built from one of CarConfig.cpp's presets (0 OCTANE, 1 DOMINUS, 2 PLANK, 3 BREAKOUT, 4 HYBRID, 5 MERC), as Gym's constructor
// does with its carConfig argument (Gym.cpp:45-49); ref_car_config: the preset's numbers, for a check of the repo's own table
static const CarConfig& preset_config(int preset) {
    const CarConfig* all[6] = { &CAR_CONFIG_OCTANE, &CAR_CONFIG_DOMINUS, &CAR_CONFIG_PLANK, &CAR_CONFIG_BREAKOUT, &CAR_CONFIG_HYBRID, &CAR_CONFIG_MERC };
    return *all[preset < 0 || preset > 5 ? 0 : preset];
}
void* ref_arena_new_cfg(int team_size, int preset) {
    Arena* a = Arena::Create(GameMode::SOCCAR);
    for (int i = 0; i < team_size; i++) { a->AddCar(Team::BLUE, preset_config(preset)); a->AddCar(Team::ORANGE, preset_config(preset)); }
    return a;
}
// ... and from any CarConfig (the 17 numbers in ref_car_config's order): the bisection tool of tests/golden/make_carconfig_golden.py's `custom` tapes
void* ref_arena_new_custom(int team_size, const float* v) {
    CarConfig c;
    c.hitboxSize = Vec(v[0], v[1], v[2]); c.hitboxPosOffset = Vec(v[3], v[4], v[5]);
    c.frontWheels.wheelRadius = v[6]; c.frontWheels.suspensionRestLength = v[7]; c.frontWheels.connectionPointOffset = Vec(v[8], v[9], v[10]);
    c.backWheels.wheelRadius = v[11]; c.backWheels.suspensionRestLength = v[12]; c.backWheels.connectionPointOffset = Vec(v[13], v[14], v[15]);
    c.dodgeDeadzone = v[16];
    Arena* a = Arena::Create(GameMode::SOCCAR);
    for (int i = 0; i < team_size; i++) { a->AddCar(Team::BLUE, c); a->AddCar(Team::ORANGE, c); }
    return a;
}
void ref_car_config(int preset, float* out17) {
    const CarConfig& c = preset_config(preset);
    const float v[17] = { c.hitboxSize.x, c.hitboxSize.y, c.hitboxSize.z, c.hitboxPosOffset.x, c.hitboxPosOffset.y, c.hitboxPosOffset.z,
                          c.frontWheels.wheelRadius, c.frontWheels.suspensionRestLength, c.frontWheels.connectionPointOffset.x, c.frontWheels.connectionPointOffset.y, c.frontWheels.connectionPointOffset.z,
                          c.backWheels.wheelRadius, c.backWheels.suspensionRestLength, c.backWheels.connectionPointOffset.x, c.backWheels.connectionPointOffset.y, c.backWheels.connectionPointOffset.z, c.dodgeDeadzone };
    memcpy(out17, v, sizeof(v));
}
// The same arena with the cars at OTHER heap addresses: Arena::_cars is a std::unordered_set<Car*>, so the order of the reference's per-car
// loops is a function of where malloc put the cars.  A few odd-sized allocations between the AddCar calls move them (kept: freeing would
// give the addresses back); callers try seeds until ref_arena_get_state reports the car order they want to reproduce (tools/raw_divergence.py).
void* ref_arena_new_shuffled(int team_size, unsigned seed) {
    Arena* a = Arena::Create(GameMode::SOCCAR);
    unsigned x = seed * 2654435761u + 12345u;
    for (int i = 0; i < 2 * team_size; i++) {
        x = x * 1664525u + 1013904223u;
        if (seed) (void)malloc(16 + (x >> 8) % 9000);
        a->AddCar((i & 1) ? Team::ORANGE : Team::BLUE);
    }
    return a;
}
// ... and with another bucket count in that set: the iteration order of the same pointers changes with it
void ref_arena_rehash(void* h, int buckets) { ((Arena*)h)->_cars.rehash((size_t)buckets); }
void ref_arena_free(void* h) { delete (Arena*)h; }
void ref_arena_get_state(void* h, RlgpuArenaState* s) { GetArenaPhys((Arena*)h, s); }
// MutatorConfig's run-time scalars (RlgpuMutators, include/rlgpu_state.h) onto the reference's arena: Arena::SetMutatorConfig (Arena.cpp:15-48) with the
// fields the stepper leaves compiled in at their defaults.  ballDrag is given as the per-tick factor's source: the caller passes the drag itself in `ball_drag`.
void ref_arena_set_mutators(void* h, const RlgpuMutators* m, float ball_drag) {
    Arena* a = (Arena*)h;
    MutatorConfig mc = a->GetMutatorConfig();
    mc.gravity = Vec(m->gravity_x, m->gravity_y, m->gravity_z);
    mc.carWorldFriction = m->car_world_friction; mc.carWorldRestitution = m->car_world_restitution; mc.ballWorldFriction = m->ball_world_friction; mc.ballWorldRestitution = m->ball_world_restitution;
    mc.boostAccelGround = m->boost_accel_ground; mc.boostAccelAir = m->boost_accel_air; mc.boostUsedPerSecond = m->boost_used_per_second;
    mc.jumpAccel = m->jump_accel; mc.jumpImmediateForce = m->jump_immediate_force;
    mc.ballMaxSpeed = m->ball_max_speed; mc.ballDrag = ball_drag;
    mc.respawnDelay = m->respawn_delay; mc.bumpCooldownTime = m->bump_cooldown_time;
    mc.boostPadCooldown_Big = m->boost_pad_cooldown_big; mc.boostPadCooldown_Small = m->boost_pad_cooldown_small;
    mc.carSpawnBoostAmount = m->car_spawn_boost_amount; mc.ballHitExtraForceScale = m->ball_hit_extra_force_scale; mc.bumpForceScale = m->bump_force_scale;
    mc.goalBaseThresholdY = m->goal_base_threshold_y;
    mc.unlimitedFlips = (m->flags & RLGPU_MUT_UNLIMITED_FLIPS) != 0; mc.unlimitedDoubleJumps = (m->flags & RLGPU_MUT_UNLIMITED_DOUBLE_JUMPS) != 0;
    mc.demoMode = (m->flags & RLGPU_MUT_DEMO_ON_CONTACT) ? DemoMode::ON_CONTACT : (m->flags & RLGPU_MUT_DEMO_DISABLED) ? DemoMode::DISABLED : DemoMode::NORMAL;
    mc.enableTeamDemos = (m->flags & RLGPU_MUT_TEAM_DEMOS) != 0;
    a->SetMutatorConfig(mc);
}
// The arena's hidden state into s->hidden (s: a state of this arena, from ref_arena_get_state): what btRSBroadphase remembers of its dynamic proxies --
// the cell each was last filed under (btRSBroadphaseProxy::cellIdx) and the order in which they last ARRIVED in their cells, which is the order of
// every cell's dynHandles list (btRSBroadphase.cpp:185-203,287-325: a proxy that changes cell is erased from its old 27 lists and pushed back onto
// the new 27) -- and the basis of every demolished car's rigid body.  The arrival order is a total order here: any order that agrees with all the
// cell lists (two proxies that share no list have no order yet; the first one to move next to the other goes to the back of the lists).
void ref_arena_get_hidden(void* h, RlgpuArenaState* s) {
    Arena* a = (Arena*)h;
    btRSBroadphase* bp = dynamic_cast<btRSBroadphase*>(a->_bulletWorld.getBroadphase());
    if (!bp) return;
    const int nb = 1 + s->num_cars;
    btRSBroadphaseProxy* px[8] = {};
    px[0] = (btRSBroadphaseProxy*)a->ball->_rigidBody.getBroadphaseHandle();
    for (int k = 0; k < s->num_cars; k++) if (Car* car = CarBySlot(a, k)) px[1 + k] = (btRSBroadphaseProxy*)car->_rigidBody.getBroadphaseHandle();
    auto idx = [&](btRSBroadphaseProxy* p) { for (int b = 0; b < nb; b++) if (px[b] == p) return b; return -1; };
    bool before[8][8] = {};
    for (auto& cell : bp->cells)
        for (size_t i = 0; i < cell.dynHandles.size(); i++)
            for (size_t j = i + 1; j < cell.dynHandles.size(); j++) {
                const int x = idx(cell.dynHandles[i]), y = idx(cell.dynHandles[j]);
                if (x >= 0 && y >= 0) before[x][y] = true;
            }
    bool placed[8] = {}; int rank[8] = {};
    for (int r = 0; r < nb; r++) {
        int pick = -1;
        for (int b = 0; b < nb && pick < 0; b++) {
            if (placed[b]) continue;
            bool free_ = true;
            for (int o = 0; o < nb; o++) if (!placed[o] && o != b && before[o][b]) free_ = false;
            if (free_) pick = b;
        }
        if (pick < 0) for (int b = 0; b < nb; b++) if (!placed[b]) { pick = b; break; }   // (cannot happen: the lists agree)
        placed[pick] = true; rank[pick] = r;
    }
    for (int b = 0; b < 8; b++) s->hidden.bp_hist[b] = 0;
    for (int b = 0; b < nb; b++) if (px[b]) s->hidden.bp_hist[b] = (uint16_t)(((uint32_t)px[b]->cellIdx << 3) | (uint32_t)rank[b]);
    for (int k = 0; k < s->num_cars; k++) {
        Car* car = CarBySlot(a, k);
        for (int q = 0; q < 9; q++) s->hidden.wreck_rot[k][q] = 0.f;
        if (!car || !car->_internalState.isDemoed) continue;
        const RotMat rm = car->_rigidBody.getWorldTransform().getBasis();
        V3(s->hidden.wreck_rot[k], rm.forward); V3(s->hidden.wreck_rot[k] + 3, rm.right); V3(s->hidden.wreck_rot[k] + 6, rm.up);
    }
    s->hidden.valid |= RLGPU_HIDDEN_BP_HIST | RLGPU_HIDDEN_WRECK_ROT;   // (the mutators bit of ref_arena_get_state stays)
}
// the proxy boxes of the arena's static planes (cap > 0) or mesh bodies (cap < 0: up to -cap) in world order: 6 floats each
int ref_debug_plane_boxes(void* h, float* out, int cap) {
    Arena* a = (Arena*)h; int n = 0; const int cap_ = cap;
    if (cap < 0) { /* mesh bodies */ }
    const btCollisionObjectArray& objs = a->_bulletWorld.getCollisionObjectArray();
    for (int i = 0; i < objs.size() && n < (cap_ < 0 ? -cap_ : cap_); i++) {
        if (objs[i]->getCollisionShape()->getShapeType() != (cap < 0 ? TRIANGLE_MESH_SHAPE_PROXYTYPE : STATIC_PLANE_PROXYTYPE)) continue;
        btBroadphaseProxy* p = objs[i]->getBroadphaseHandle();
        for (int k = 0; k < 3; k++) { out[6 * n + k] = p->m_aabbMin[k]; out[6 * n + 3 + k] = p->m_aabbMax[k]; }
        n++;
    }
    return n;
}
// the broadphase's box of a dynamic body as the last setAabb left it (body 0: the ball, 1 + k: car slot k): min[3], max[3]
void ref_debug_proxy_aabb(void* h, int body, float* out6) {
    Arena* a = (Arena*)h;
    btBroadphaseProxy* p = body == 0 ? a->ball->_rigidBody.getBroadphaseHandle() : CarBySlot(a, body - 1)->_rigidBody.getBroadphaseHandle();
    for (int k = 0; k < 3; k++) { out6[k] = p->m_aabbMin[k]; out6[3 + k] = p->m_aabbMax[k]; }
}
void ref_arena_set_state(void* h, const RlgpuArenaState* s) { SetArenaPhys((Arena*)h, s, true); }
void ref_arena_set_controls(void* h, int slot, const float* c8) { CarBySlot((Arena*)h, slot)->controls = ArrToCtrl(c8); }
void ref_arena_step(void* h, int ticks) { ((Arena*)h)->Step(ticks); }
void ref_arena_reset_kickoff(void* h, int seed) { ((Arena*)h)->ResetToRandomKickoff(seed); }

// The 90x8 lookup table of DiscreteAction (DiscreteAction.cpp:3-67).
int ref_action_table(float* out, int cap_rows) {
    DiscreteAction da;
    int n = (int)da.actions.size();
    for (int i = 0; i < n && i < cap_rows; i++) for (int k = 0; k < 8; k++) out[i * 8 + k] = da.actions[i][k];
    return n;
}

// reward_kind 0: examplemain.cpp:62-76 stack. 1: same wrapped in ZeroSumReward(teamSpirit 0.5, oppScale 1).
// obs_kind 0: DefaultOBS. terminal: NoTouchCondition(no_touch_steps) then GoalScoreCondition (examplemain.cpp:78-81).
void* ref_gym_new(int team_size, int tick_skip, int obs_kind, int reward_kind, int no_touch_steps) {
    RefGym* g = new RefGym();
    g->eventReward = new EventReward({ .teamGoal = 1.f, .concede = -1.f });
    auto* comb = new CombinedReward({
        { new FaceBallReward(), 0.1f }, { new VelocityPlayerToBallReward(), 0.5f },
        { new VelocityBallToGoalReward(), 1.0f }, { g->eventReward, 50.f } });
    g->rootReward = comb;
    if (reward_kind == 1) g->rootReward = new ZeroSumReward(comb, 0.5f, 1.0f);
    g->noTouch = new NoTouchCondition(no_touch_steps);
    g->conds = { g->noTouch, new GoalScoreCondition() };
    g->obs = new DefaultOBS();
    g->parser = new DiscreteAction();
    g->setter = new FixedStateSetter();
    g->match = new Match(g->rootReward, g->conds, g->obs, g->parser, g->setter, team_size, true);
    g->gym = new Gym(g->match, tick_skip);
    g->nPlayers = team_size * 2;
    return g;
}
void ref_gym_free(void* h) { RefGym* g = (RefGym*)h; delete g->gym; delete g->match; delete g; }

void* ref_gym_arena(void* h) { return ((RefGym*)h)->gym->arena; }

// Reset to a caller-provided physical state (pads are reset to default by Match::ResetState, Match.cpp:67-68).
// obs_out: [nPlayers x D] in SLOT order. Returns D.
int ref_gym_reset_to(void* h, const RlgpuArenaState* s, float* obs_out) {
    RefGym* g = (RefGym*)h;
    g->setter->next = *s;
    FList2 obs = g->gym->Reset();
    int D = (int)obs[0].size();
    const auto& pl = g->gym->prevState.players;
    for (size_t i = 0; i < pl.size(); i++) memcpy(obs_out + ((int)pl[i].carId - 1) * D, obs[i].data(), D * 4);
    return D;
}

// One Gym::Step with SLOT-ordered action indices. Outputs in SLOT order.
//  (rows are AGENT rows: slot order with two teams; blue car order in a one-team gym, where agent row = car id - 1 = slot / 2)
//  snap_out (optional): the physical state at the snapshot (after tick 1 of tickSkip)
//  is not observable without modifying the reference; state_out is the arena AFTER the full step,
//  with gym-level carried state filled in.
void ref_gym_step(void* h, const int32_t* actions, float* obs_out, float* rew_out, int32_t* done_out, RlgpuArenaState* state_out) {
    RefGym* g = (RefGym*)h;
    Arena* a = g->gym->arena;
    IList in(g->nPlayers);
    int i = 0;
    for (Car* c : a->_cars) in[i++] = actions[(int)c->id - 1];
    Gym::StepResult r = g->gym->Step(in);
    int D = (int)r.obs[0].size();
    for (size_t k = 0; k < r.state.players.size(); k++) {
        int slot = (int)r.state.players[k].carId - 1;
        memcpy(obs_out + slot * D, r.obs[k].data(), D * 4);
        rew_out[slot] = r.reward[k];
    }
    *done_out = r.done;
    if (state_out) { GetArenaPhys(a, state_out); FillGymState(g, state_out); }
}

// ---- CPU baseline: the reference's own stepping path, threads x games as in ThreadAgent -------------
// Steps `n_envs` 1v1 gyms (example obs/reward/terminal stack, RandomState(true,true,true), uniform
// random action tape, tickSkip 8) for `steps` gym steps each on `n_threads` threads.  Returns seconds.
double ref_bench_collect(int team_size, int n_envs, int n_threads, int steps, int tick_skip) {
    struct Env { Match* m; Gym* g; };
    std::vector<Env> envs(n_envs);
    for (auto& e : envs) {
        auto* rew = new CombinedReward({ { new FaceBallReward(), 0.1f }, { new VelocityPlayerToBallReward(), 0.5f },
            { new VelocityBallToGoalReward(), 1.0f }, { new EventReward({ .teamGoal = 1.f, .concede = -1.f }), 50.f } });
        std::vector<TerminalCondition*> tc = { new NoTouchCondition(150), new GoalScoreCondition() };
        e.m = new Match(rew, tc, new DefaultOBS(), new DiscreteAction(), new RandomState(true, true, true), team_size, true);
        e.g = new Gym(e.m, tick_skip);
        e.g->Reset();
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) {
        th.emplace_back([&, t]() {
            uint32_t rng = 12345u + 977u * t;
            for (int s = 0; s < steps; s++)
                for (int e = t; e < n_envs; e += n_threads) {
                    IList acts(team_size * 2);
                    for (auto& a : acts) { rng = rng * 1664525u + 1013904223u; a = (rng >> 8) % 90; }
                    auto r = envs[e].g->Step(acts);
                    if (r.done) envs[e].g->Reset();
                }
        });
    }
    for (auto& x : th) x.join();
    double sec = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    for (auto& e : envs) { delete e.g; delete e.m; }
    return sec;
}

}  // extern "C"

// ---- debugging probes (used while pinning the port; not part of any test contract) ----------------------
extern "C" void ref_debug_wheels(void* h, int slot, float* out /*4 x 12*/) {
    Car* car = CarBySlot((Arena*)h, slot);
    for (int w = 0; w < 4; w++) {
        const auto& wi = car->_bulletVehicle.m_wheelInfo[w];
        float* o = out + w * 12;
        o[0] = wi.m_raycastInfo.m_suspensionLength; o[1] = wi.m_wheelsSuspensionForce; o[2] = wi.m_suspensionRelativeVelocity;
        o[3] = wi.m_clippedInvContactDotSuspension;
        o[4] = wi.m_raycastInfo.m_contactPointWS.x(); o[5] = wi.m_raycastInfo.m_contactPointWS.y(); o[6] = wi.m_raycastInfo.m_contactPointWS.z();
        o[7] = wi.m_raycastInfo.m_contactNormalWS.x(); o[8] = wi.m_raycastInfo.m_contactNormalWS.y(); o[9] = wi.m_raycastInfo.m_contactNormalWS.z();
        o[10] = wi.m_raycastInfo.m_hardPointWS.z(); o[11] = wi.m_impulse.length();
    }
}
extern "C" void ref_probe_thresholds(void* h, float* out /*4*/) {
    Arena* a = (Arena*)h;
    Car* car = CarBySlot(a, 0);
    out[0] = a->ball->_rigidBody.getCollisionShape()->getContactBreakingThreshold(gContactBreakingThreshold);
    out[1] = car->_rigidBody.getCollisionShape()->getContactBreakingThreshold(gContactBreakingThreshold);
    btVector3 inertia = car->_rigidBody.getLocalInertia();
    out[2] = inertia.x(); out[3] = inertia.y(); out[4] = inertia.z();
    out[5] = a->ball->_rigidBody.getLocalInertia().x();
}

// Dump of the dispatcher's contact manifolds as they stand after the last Arena::Step (they live until the next tick's
// broadphase pass, which removes every pair: btRSBroadphase.cpp calculateOverlappingPairs).  Per point 16 floats:
//   [0] body0 kind (0 ball, 1+slot car, -1 static)  [1] body1 kind  [2] manifold index  [3] lifetime
//   [4..6] positionWorldOnA  [7..9] positionWorldOnB  [10..12] normalWorldOnB  [13] distance  [14] appliedImpulse  [15] isSpecial
static float BodyKind(const btCollisionObject* o) {
    if (o->getUserIndex() == BT_USERINFO_TYPE_BALL) return 0.f;
    if (o->getUserIndex() == BT_USERINFO_TYPE_CAR) return (float)((Car*)o->getUserPointer())->id;
    return -1.f;
}
extern "C" int ref_debug_manifolds(void* h, float* out, int cap_points) {
    Arena* a = (Arena*)h;
    btCollisionDispatcher* d = (btCollisionDispatcher*)a->_bulletWorld.getDispatcher();
    int n = 0;
    for (int m = 0; m < d->getNumManifolds(); m++) {
        btPersistentManifold* pm = d->getManifoldByIndexInternal(m);
        for (int p = 0; p < pm->getNumContacts() && n < cap_points; p++) {
            const btManifoldPoint& cp = pm->getContactPoint(p);
            float* o = out + n * 16;
            o[0] = BodyKind(pm->getBody0()); o[1] = BodyKind(pm->getBody1()); o[2] = (float)m; o[3] = (float)cp.m_lifeTime;
            for (int k = 0; k < 3; k++) { o[4 + k] = cp.m_positionWorldOnA[k]; o[7 + k] = cp.m_positionWorldOnB[k]; o[10 + k] = cp.m_normalWorldOnB[k]; }
            o[13] = cp.m_distance1; o[14] = cp.m_appliedImpulse; o[15] = cp.m_isSpecial ? 1.f : 0.f;
            n++;
        }
    }
    return n;
}

// The order in which the solver visited the manifolds of the last tick: btSimulationIslandManager::m_islandmanifold as buildAndProcessIslands sorted it and handed it, island
// by island, to the solver (btSimulationIslandManager.cpp:196-450).  The member is private; the explicit-instantiation idiom below reads it without touching the header.
// Per manifold 2 floats: body0 kind, body1 kind (BodyKind above).
namespace {
struct IslandManifoldsTag { typedef btAlignedObjectArray<btPersistentManifold*> btSimulationIslandManager::*type; friend type get(IslandManifoldsTag); };
template <class Tag, typename Tag::type M> struct PrivateMember { friend typename Tag::type get(Tag) { return M; } };
template struct PrivateMember<IslandManifoldsTag, &btSimulationIslandManager::m_islandmanifold>;
}
extern "C" int ref_debug_island_order(void* h, float* out, int cap) {
    Arena* a = (Arena*)h;
    btSimulationIslandManager* mgr = a->_bulletWorld.getSimulationIslandManager();
    btAlignedObjectArray<btPersistentManifold*>& arr = mgr->*get(IslandManifoldsTag());
    int n = 0;
    for (int i = 0; i < arr.size() && n < cap; i++) { out[2 * n] = BodyKind(arr[i]->getBody0()); out[2 * n + 1] = BodyKind(arr[i]->getBody1()); n++; }
    return n;
}

// friction side of the same points, in the same order: 8 floats = lateralFrictionDir1[3], appliedImpulseLateral1, combinedFriction, combinedRestitution, appliedImpulse, 0
extern "C" int ref_debug_manifold_friction(void* h, float* out, int cap_points) {
    Arena* a = (Arena*)h;
    btCollisionDispatcher* d = (btCollisionDispatcher*)a->_bulletWorld.getDispatcher();
    int n = 0;
    for (int m = 0; m < d->getNumManifolds(); m++) {
        btPersistentManifold* pm = d->getManifoldByIndexInternal(m);
        for (int p = 0; p < pm->getNumContacts() && n < cap_points; p++) {
            const btManifoldPoint& cp = pm->getContactPoint(p);
            float* o = out + n * 8;
            for (int k = 0; k < 3; k++) o[k] = cp.m_lateralFrictionDir1[k];
            o[3] = cp.m_appliedImpulseLateral1; o[4] = cp.m_combinedFriction; o[5] = cp.m_combinedRestitution; o[6] = cp.m_appliedImpulse; o[7] = 0.f;
            n++;
        }
    }
    return n;
}

// every manifold of the dispatcher in array order, points or not: 4 floats each = body0 kind, body1 kind, contacts, island tag of body0
extern "C" int ref_debug_manifold_list(void* h, float* out, int cap) {
    Arena* a = (Arena*)h;
    btCollisionDispatcher* d = (btCollisionDispatcher*)a->_bulletWorld.getDispatcher();
    int n = 0;
    for (int m = 0; m < d->getNumManifolds() && n < cap; m++) {
        btPersistentManifold* pm = d->getManifoldByIndexInternal(m);
        float* o = out + 4 * n++;
        o[0] = BodyKind(pm->getBody0()); o[1] = BodyKind(pm->getBody1()); o[2] = (float)pm->getNumContacts(); o[3] = (float)pm->getBody0()->getIslandTag();
        if (o[1] < 0) {   // which static: plane normals / mesh
            const btCollisionShape* s = pm->getBody1()->getCollisionShape();
            if (s->getShapeType() == STATIC_PLANE_PROXYTYPE) { const btVector3& pn = ((const btStaticPlaneShape*)s)->getPlaneNormal(); o[1] = -(2.f + (pn.z() > 0.5f ? 0.f : pn.z() < -0.5f ? 1.f : pn.x() > 0.5f ? 2.f : 3.f)); }
        }
    }
    return n;
}

// ---- round 2: gyms beyond the example stack -----------------------------------------------------------
// obs_max_players 0: DefaultOBS, m > 0: DefaultOBSPadded(m).  reward_kind 0 / 1: the example stack plain / inside ZeroSumReward(0.5, 1);
// 2 / 3: EVERY CommonRewards.h term with distinct weights -- EventReward with eleven distinct weights, VelocityReward(false),
// SaveBoostReward(0.5), VelocityBallToGoalReward(false), VelocityPlayerToBallReward, FaceBallReward, TouchBallReward(0.7) -- plain /
// inside ZeroSumReward(0.3, 0.8).  The weights are mirrored by tests/simlib.py:all_terms_cfg.
extern "C" void* ref_gym_new2(int team_size, int tick_skip, int obs_max_players, int reward_kind, int no_touch_steps) {
    RefGym* g = new RefGym();
    CombinedReward* comb;
    if (reward_kind < 2) {
        g->eventReward = new EventReward({ .teamGoal = 1.f, .concede = -1.f });
        comb = new CombinedReward({ { new FaceBallReward(), 0.1f }, { new VelocityPlayerToBallReward(), 0.5f },
            { new VelocityBallToGoalReward(), 1.0f }, { g->eventReward, 50.f } });
    } else {
        g->eventReward = new EventReward({ .goal = 1.f, .teamGoal = 0.5f, .concede = -0.75f, .assist = 0.6f, .touch = 0.05f, .shot = 0.3f,
            .shotPass = 0.2f, .save = 0.4f, .demo = 0.35f, .demoed = -0.25f, .boostPickup = 0.15f });
        comb = new CombinedReward({ { g->eventReward, 10.f }, { new VelocityReward(false), 0.11f }, { new SaveBoostReward(0.5f), 0.07f },
            { new VelocityBallToGoalReward(false), 0.9f }, { new VelocityPlayerToBallReward(), 0.45f }, { new FaceBallReward(), 0.13f },
            { new TouchBallReward(0.7f), 0.8f } });
    }
    g->rootReward = comb;
    if (reward_kind == 1) g->rootReward = new ZeroSumReward(comb, 0.5f, 1.0f);
    if (reward_kind == 3) g->rootReward = new ZeroSumReward(comb, 0.3f, 0.8f);
    g->noTouch = new NoTouchCondition(no_touch_steps);
    g->conds = { g->noTouch, new GoalScoreCondition() };
    g->obs = obs_max_players > 0 ? (OBSBuilder*)new DefaultOBSPadded(obs_max_players) : (OBSBuilder*)new DefaultOBS();
    g->parser = new DiscreteAction();
    g->setter = new FixedStateSetter();
    g->match = new Match(g->rootReward, g->conds, g->obs, g->parser, g->setter, team_size, true);
    g->gym = new Gym(g->match, tick_skip);
    g->nPlayers = team_size * 2;
    return g;
}
// the same gyms without opponents: Match(..., teamSize, spawnOpponents = false)
extern "C" void* ref_gym_new3(int team_size, int tick_skip, int obs_max_players, int reward_kind, int no_touch_steps, int spawn_opponents) {
    RefGym* g = (RefGym*)ref_gym_new2(team_size, tick_skip, obs_max_players, reward_kind, no_touch_steps);
    if (spawn_opponents) return g;
    delete g->gym; delete g->match;
    g->match = new Match(g->rootReward, g->conds, g->obs, g->parser, g->setter, team_size, false);
    g->gym = new Gym(g->match, tick_skip);
    g->nPlayers = team_size;
    return g;
}
// car ids (slot + 1) in the order of GameState::players, i.e. the reference's own iteration order of Arena::_cars (a
// std::unordered_set<Car*>: pointer-hash order, different from run to run) -- DefaultOBS lists the other players in it
extern "C" int ref_gym_player_order(void* h, int32_t* out) {
    RefGym* g = (RefGym*)h;
    int n = 0;
    for (auto& p : g->gym->prevState.players) out[n++] = (int32_t)p.carId;
    return n;
}
// ---- round 4: GameInst::Step across episode ends (PUB/Threading/GameInst.cpp:7-38, compiled unedited next to this file: oracle/Makefile) ----
// A user state setter written against the reference's plugin surface: its k-th call installs the caller's k-th state (the last one again once
// the list is used up).  With it an episode boundary -- gym->Reset() inside GameInst::Step, the observation the agent acts on next, the reward
// trackers' roll-over -- is reproducible, which the reference's own setters (thread-local std RNG) are not.
// (ref_list_setter_then_random(1): once the list is used up the calls go to the reference's own RandomState instead -- the first episode starts
// from a chosen state, every later one the way the example program's setter starts it; tests/golden/make_padreset_golden.py)
static int g_list_then_random = 0;
extern "C" void ref_list_setter_then_random(int on) { g_list_then_random = on; }
class ListStateSetter : public StateSetter {
public:
    const RlgpuArenaState* list = nullptr; int n = 0, calls = 0;
    RandomState random{true, true, true};
    virtual GameState ResetState(Arena* arena) {
        if (g_list_then_random && calls >= n) { calls++; return random.ResetState(arena); }
        SetArenaPhys(arena, &list[calls < n ? calls : n - 1], false);
        calls++;
        return GameState(arena);
    }
};
// Runs GameInst::Start() and n_steps x GameInst::Step(actions[t]) on one of the gyms above.  actions [n_steps][players] by SLOT.  Outputs by slot:
// cur_obs_out [(n_steps + 1)][players][D] = GameInst::curObs after Start() and after every Step (the NEW episode's first observation when the step
// ended one); step_obs_out [n_steps][players][D] = StepResult::obs (the same rows: GameInst overwrites them before it returns); rew_out [n_steps][players];
// (both in the players' order AFTER the step -- order_out -- which is what the rows' "other players" blocks follow); done_out [n_steps]; trackers_out [n_steps][6] = curEpRew, avgEpRew.total, avgEpRew.count, avgStepRew.total, avgStepRew.count, totalSteps;
// resets_out [n_steps + 1] = state-setter calls made so far.  Returns D.
extern "C" int ref_gameinst_run(int team_size, int tick_skip, int obs_max_players, int reward_kind, int no_touch_steps, const RlgpuArenaState* states, int n_states,
                                const int32_t* actions, int n_steps, float* cur_obs_out, float* step_obs_out, float* rew_out, int32_t* done_out, float* trackers_out, int32_t* resets_out,
                                int32_t* order_out /* [(n_steps + 1)][players]: car ids - 1 in the order of the gym's current GameState::players */) {
    RefGym* g = (RefGym*)ref_gym_new2(team_size, tick_skip, obs_max_players, reward_kind, no_touch_steps);
    ListStateSetter* ls = new ListStateSetter(); ls->list = states; ls->n = n_states;
    g->match->stateSetter = ls;
    const int P = g->nPlayers;
    int D = 0;
    {
        RLGPC::GameInst gi(g->gym, g->match);   // (owns and deletes the gym and the match, GameInst.h:53-56)
        gi.Start();
        D = (int)gi.curObs[0].size();
        auto put_obs = [&](float* dst, const FList2& obs) {
            const auto& pl = g->gym->prevState.players;
            for (size_t i = 0; i < pl.size(); i++) memcpy(dst + ((int)pl[i].carId - 1) * D, obs[i].data(), (size_t)D * 4);
        };
        auto put_order = [&](int32_t* dst) { int n = 0; for (auto& p : g->gym->prevState.players) dst[n++] = (int32_t)p.carId - 1; };
        put_obs(cur_obs_out, gi.curObs);
        put_order(order_out);
        resets_out[0] = ls->calls;
        for (int t = 0; t < n_steps; t++) {
            IList in(P);
            int i = 0;
            for (Car* c : g->gym->arena->_cars) in[i++] = actions[(size_t)t * P + (int)c->id - 1];
            Gym::StepResult r = gi.Step(in);
            // rewards belong to the players of the state the step was made in (r.state), the observations to the state the gym holds now
            for (size_t k = 0; k < r.state.players.size(); k++) rew_out[(size_t)t * P + (int)r.state.players[k].carId - 1] = r.reward[k];
            put_obs(step_obs_out + (size_t)t * P * D, r.obs);
            put_obs(cur_obs_out + (size_t)(t + 1) * P * D, gi.curObs);
            done_out[t] = r.done;
            float* tr = trackers_out + (size_t)t * 6;
            tr[0] = gi.curEpRew; tr[1] = gi.avgEpRew.total; tr[2] = (float)gi.avgEpRew.count; tr[3] = gi.avgStepRew.total; tr[4] = (float)gi.avgStepRew.count; tr[5] = (float)gi.totalSteps;
            resets_out[t + 1] = ls->calls;
            put_order(order_out + (size_t)(t + 1) * P);
        }
    }
    delete ls; delete g;
    return D;
}

// state setters: `n` arenas reset by the reference's own RandomState(ballRandSpeed, carRandSpeed, carsOnGround) (kind 0) or KickoffState (kind 1)
extern "C" void ref_setter_samples(int team_size, int kind, int n, RlgpuArenaState* out) {
    Arena* a = Arena::Create(GameMode::SOCCAR);
    for (int i = 0; i < team_size; i++) { a->AddCar(Team::BLUE); a->AddCar(Team::ORANGE); }
    RandomState rs(true, true, true); KickoffState ks;
    for (int i = 0; i < n; i++) {
        if (kind == 0) rs.ResetState(a); else ks.ResetState(a);
        GetArenaPhys(a, &out[i]);
    }
    delete a;
}

// ---- the reference's random engine, pinned (round 6) -------------------------------------------------------------------------------------------
// Math::GetRandEngine() hands out a REFERENCE to the calling thread's std::default_random_engine (RocketSim Math.cpp:59-64), which the reference seeds
// from the wall clock.  Assigning it an engine in a known state makes every draw this thread makes afterwards -- Car::Respawn's slot (Car.cpp:48),
// ResetToRandomKickoff's shuffle (Arena.cpp:127-134), RandomState's values (RandomState.cpp:8-61), DefaultOBSPadded's shuffles -- a function of that
// state; the stepper's test mode (RlgpuArenaHidden::ref_engine) draws from the same state with the same formulas, so both sides can be compared for
// equality through respawns and resets.  The engine's state is its last output (minstd_rand0).
#include <sstream>
// btCollisionWorld::rayTest with a ClosestRayResultCallback -- what btDefaultVehicleRaycaster::castRay runs for a wheel (btDefaultVehicleRaycaster.cpp:20-51) -- against
// the arena's static world (park the ball and the cars elsewhere): out4 = closest hit fraction, world normal; returns 1 on a hit.  from / to in Bullet units.
extern "C" int ref_ray_world(void* h, const float* from3, const float* to3, float* out4) {
    Arena* a = (Arena*)h;
    const btVector3 from(from3[0], from3[1], from3[2]), to(to3[0], to3[1], to3[2]);
    btCollisionWorld::ClosestRayResultCallback cb(from, to, nullptr);
    a->_bulletWorld.rayTest(from, to, cb);
    out4[0] = cb.m_closestHitFraction; out4[1] = cb.m_hitNormalWorld.x(); out4[2] = cb.m_hitNormalWorld.y(); out4[3] = cb.m_hitNormalWorld.z();
    return cb.hasHit() ? 1 : 0;
}
extern "C" void ref_seed_engine(unsigned state) { RocketSim::Math::GetRandEngine() = std::default_random_engine(state); }
extern "C" unsigned ref_engine_state() { std::ostringstream os; os << RocketSim::Math::GetRandEngine(); return (unsigned)std::stoul(os.str()); }
// `n` resets of one arena by the reference's own setters with the thread's engine started from `engine`: kind 0 RandomState(flags bit 0 ball speed, bit 1
// car speed, bit 2 cars on the ground), kind 1 KickoffState (the thread's engine), kind 2 Arena::ResetToRandomKickoff(seed0 + i) (an engine of its own per
// call: Arena.cpp:126-131).  `rehash` > 0 re-buckets the arena's car set first (another iteration order of `_cars`).  out[i] = the state after reset i
// (car_order included), engine_after[i] = the thread engine's state after it.
extern "C" void ref_setter_samples_seeded(int team_size, int kind, int flags, int n, unsigned engine, int seed0, int rehash, RlgpuArenaState* out, unsigned* engine_after) {
    Arena* a = Arena::Create(GameMode::SOCCAR);
    for (int i = 0; i < team_size; i++) { a->AddCar(Team::BLUE); a->AddCar(Team::ORANGE); }
    if (rehash > 0) a->_cars.rehash((size_t)rehash);
    RandomState rs((flags & 1) != 0, (flags & 2) != 0, (flags & 4) != 0); KickoffState ks;
    ref_seed_engine(engine);
    for (int i = 0; i < n; i++) {
        if (kind == 0) rs.ResetState(a); else if (kind == 1) ks.ResetState(a); else a->ResetToRandomKickoff(seed0 + i);
        GetArenaPhys(a, &out[i]);
        engine_after[i] = ref_engine_state();
    }
    delete a;
}

// Bullet-unit state of the ball and of every car slot, straight from the rigid bodies (no unit conversion, no rounding): per body 18 floats
// = origin[3], basis rows[9], linear velocity[3], angular velocity[3].  out: (1 + n_slots) x 18.  For tools/raw_divergence.py, which looks
// for differences the uu exchange format rounds away.
extern "C" void ref_arena_get_raw(void* h, int n_slots, float* out) {
    Arena* a = (Arena*)h;
    auto put = [&](float* o, const btRigidBody& rb) {
        const btTransform& t = rb.getWorldTransform();
        for (int i = 0; i < 3; i++) o[i] = t.getOrigin()[i];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) o[3 + 3 * r + c] = t.getBasis()[r][c];
        for (int i = 0; i < 3; i++) { o[12 + i] = rb.getLinearVelocity()[i]; o[15 + i] = rb.getAngularVelocity()[i]; }
    };
    put(out, a->ball->_rigidBody);
    for (int k = 0; k < n_slots; k++) {
        Car* car = CarBySlot(a, k);
        float* o = out + 18 * (1 + k);
        if (car) put(o, car->_rigidBody); else for (int i = 0; i < 18; i++) o[i] = 0.f;
    }
}
