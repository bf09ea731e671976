// GameState / PlayerData / PhysObj / ScoreLine (SIM/Utils/Gamestates/{GameState.h:6-57, PlayerData.h:7-38, PhysObj.h}).
// Filled from the device state (rlgpu_env_download_states) when a step callback asks for it -- the slow path.
#pragma once
#include "../../Framework.h"
#include "../BasicTypes/Action.h"
struct RlgpuArenaState;
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
    // materialise from one downloaded env (rlgymppo_cpp_amd/host/Host.cpp)
    explicit GameState(const RlgpuArenaState& s, int tickSkip);
    const PhysObj& GetBallPhys(bool inverted) const { return inverted ? ballInv : ball; }
    const bool* GetBoostPads(bool inverted) const { return inverted ? boostPadsInv : boostPads; }
};
}
