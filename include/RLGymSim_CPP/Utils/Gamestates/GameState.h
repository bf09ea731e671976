// GameState / PlayerData / PhysObj / ScoreLine (SIM/Utils/Gamestates/{GameState.h:6-57, PlayerData.h:7-38, PhysObj.h}).
// Built from an env's state in the exchange layout of rlgpu_state.h: the snapshot the step kernel stores for host plugins and step
// callbacks (rlgpu_env_enable_snapshots), or the state inside a host Arena facade.
#pragma once
#include "../../Framework.h"
#include "../../RocketSim/Arena.h"
#include "../BasicTypes/Action.h"
namespace RLGSC {
struct PhysObj {
    Vec pos, vel, angVel; RotMat rotMat;
    PhysObj Invert() const {  // PhysObj.cpp:19-31: x and y mirrored
        PhysObj r = *this; const Vec s(-1, -1, 1);
        r.pos = pos * s; r.vel = vel * s; r.angVel = angVel * s;
        r.rotMat.forward = rotMat.forward * s; r.rotMat.right = rotMat.right * s; r.rotMat.up = rotMat.up * s;
        return r;
    }
};
struct PlayerData {
    uint32_t carId = 0; Team team = Team::BLUE;
    PhysObj phys, physInv; CarState carState;
    int matchGoals = 0, matchSaves = 0, matchAssists = 0, matchShots = 0, matchShotPasses = 0, matchBumps = 0, matchDemos = 0, boostPickups = 0;
    bool hasJump = true, hasFlip = true;
    float boostFraction = 0.33f;
    bool ballTouchedStep = false, ballTouchedTick = false;
    const PhysObj& GetPhys(bool inverted) const { return inverted ? physInv : phys; }
};
struct ScoreLine { int teamGoals[2] = {0, 0}; int operator[](size_t i) const { return teamGoals[i]; } };
struct GameState {
    ScoreLine scoreLine; int lastTouchCarID = -1;
    std::vector<PlayerData> players;
    PhysObj ball, ballInv;
    bool boostPads[CommonValues::BOOST_LOCATIONS_AMOUNT] = {}, boostPadsInv[CommonValues::BOOST_LOCATIONS_AMOUNT] = {};
    uint64_t lastTickCount = 0; int deltaTickCount = 0;
    GameState() = default;
    // from one env's state; deltaTicks = ticks since the previous GameState of that env (the window of PlayerData::ballTouchedStep)
    explicit GameState(const RlgpuArenaState& s, int deltaTicks);
    // GameState(Arena*) / UpdateFromArena (GameState.cpp:52-104) on the host facade: physics, pads, flags from the arena; the match
    // counters and the score line are the ones the device step keeps in the arena's gym block
    explicit GameState(Arena* arena) { UpdateFromArena(arena); }
    void UpdateFromArena(Arena* arena) {
        const uint64_t before = lastTickCount;
        arena->_SyncToState();
        *this = GameState(arena->_state, (int)(arena->tickCount - before));
    }
    const PhysObj& GetBallPhys(bool inverted) const { return inverted ? ballInv : ball; }
    const bool* GetBoostPads(bool inverted) const { return inverted ? boostPadsInv : boostPads; }
};
// GameState::UpdateFromArena / PlayerData::UpdateFromCar (SIM/Utils/Gamestates/GameState.cpp:52-104, PlayerData.cpp:4-34) from a
// state in the exchange layout
inline GameState::GameState(const RlgpuArenaState& s, int tickSkip) {
    auto V = [](const float* p) { return Vec(p[0], p[1], p[2]); };
    scoreLine.teamGoals[0] = s.gym.score_line[0]; scoreLine.teamGoals[1] = s.gym.score_line[1];
    lastTouchCarID = s.gym.last_touch_car_id;
    lastTickCount = (uint64_t)s.tick_count; deltaTickCount = tickSkip;
    ball.pos = V(s.ball.pos); ball.vel = V(s.ball.vel); ball.angVel = V(s.ball.ang_vel);
    ball.rotMat.forward = V(s.hidden.ball_rot); ball.rotMat.right = V(s.hidden.ball_rot + 3); ball.rotMat.up = V(s.hidden.ball_rot + 6);   // PhysObj(BallState): PhysObj.cpp:6
    ballInv = ball.Invert();
    // RLGym pad order -> RocketSim pad index: the map GameState.cpp:10-50 builds by matching CommonValues::BOOST_LOCATIONS against
    // the arena's pads (a constant of the two tables; the device obs builder uses the same one)
    static const int8_t PAD_ORDER[RLGPU_NUM_PADS] = {6, 7, 8, 4, 5, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 0, 19, 20, 1, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 2, 3, 31, 32, 33};
    for (int p = 0; p < RLGPU_NUM_PADS; p++) { boostPads[p] = s.pads[PAD_ORDER[p]].is_active != 0; boostPadsInv[RLGPU_NUM_PADS - 1 - p] = boostPads[p]; }
    // a one-team env (spawnOpponents = false) keeps its orange slots empty: its players are the even slots, numbered 1, 2, 3 like the
    // reference's only cars
    bool oneTeam = false;
    for (int k = 1; k < s.num_cars; k += 2) oneTeam = oneTeam || (s.cars[k].flags & RLGPU_CF_ABSENT);
    if (oneTeam && lastTouchCarID > 0) lastTouchCarID = (lastTouchCarID - 1) / 2 + 1;
    players.clear();
    for (int k = 0; k < s.num_cars; k++) {
        const RlgpuCarState& c = s.cars[k]; const RlgpuPlayerGymState& g = s.gym.players[k];
        if (c.flags & RLGPU_CF_ABSENT) continue;
        players.emplace_back();
        PlayerData& pd = players.back();
        pd.carId = oneTeam ? (uint32_t)(k / 2 + 1) : (uint32_t)(k + 1); pd.team = (k % 2 == 0) ? Team::BLUE : Team::ORANGE;
        pd.phys.pos = V(c.pos); pd.phys.vel = V(c.vel); pd.phys.angVel = V(c.ang_vel);
        pd.phys.rotMat.forward = V(c.rot); pd.phys.rotMat.right = V(c.rot + 3); pd.phys.rotMat.up = V(c.rot + 6);
        pd.physInv = pd.phys.Invert();
        CarState& cs = pd.carState;
        cs.pos = pd.phys.pos; cs.vel = pd.phys.vel; cs.angVel = pd.phys.angVel; cs.rotMat = pd.phys.rotMat;
        cs.isOnGround = c.flags & RLGPU_CF_ON_GROUND; cs.hasJumped = c.flags & RLGPU_CF_HAS_JUMPED; cs.hasDoubleJumped = c.flags & RLGPU_CF_HAS_DOUBLE_JUMPED;
        cs.hasFlipped = c.flags & RLGPU_CF_HAS_FLIPPED; cs.isJumping = c.flags & RLGPU_CF_IS_JUMPING; cs.isFlipping = c.flags & RLGPU_CF_IS_FLIPPING;
        cs.isSupersonic = c.flags & RLGPU_CF_IS_SUPERSONIC; cs.isDemoed = c.flags & RLGPU_CF_IS_DEMOED;
        cs.boost = c.boost; cs.airTimeSinceJump = c.air_time_since_jump; cs.jumpTime = c.jump_time; cs.flipTime = c.flip_time; cs.demoRespawnTimer = c.demo_respawn_timer;
        pd.matchGoals = g.match_goals; pd.matchSaves = g.match_saves; pd.matchAssists = g.match_assists; pd.matchShots = g.match_shots;
        pd.matchShotPasses = g.match_shot_passes; pd.matchBumps = g.match_bumps; pd.matchDemos = g.match_demos; pd.boostPickups = g.boost_pickups;
        pd.boostFraction = c.boost / 100.f;
        // PlayerData.cpp:20-30
        pd.ballTouchedStep = (c.flags & RLGPU_CF_BALLHIT_VALID) && c.bh_tick_hit >= s.tick_count - tickSkip;
        pd.ballTouchedTick = (c.flags & RLGPU_CF_BALLHIT_VALID) && c.bh_tick_hit == s.tick_count - 1;
        pd.hasJump = !cs.hasJumped;
        pd.hasFlip = !cs.hasDoubleJumped && !cs.hasFlipped && cs.airTimeSinceJump < 1.25f;   // RLConst::DOUBLEJUMP_MAX_DELAY
    }
}
}
