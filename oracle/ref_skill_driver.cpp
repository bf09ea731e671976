// oracle/ref_skill_driver.cpp -- TEST INFRASTRUCTURE ONLY (never linked into, imported by or executed from the product).
// A C-ABI door into the REAL reference skill tracker: PRIV/Util/SkillTracker.cpp, PUB/Threading/GameInst.cpp, PUB/Util/RenderSender.cpp,
// PRIV/PPO/DiscretePolicy.cpp compiled unedited from where they lie under /root/reference (oracle/Makefile, target ref_skill) against the
// torch wheel's libtorch, the image's pybind11 / libpython (RenderSender.h wants them; no render sender is ever created here) and the
// reference simulator archive oracle/_ref/libref_sim.a.  What this file adds: argument marshalling, and one user plugin written against
// the reference's own headers (a state setter that starts every episode with the ball behind a goal line) for the scripted games.
// tests/golden/make_skill_golden.py runs it and commits what it returns as tests/golden/skill_golden.json.
#include <private/RLGymPPO_CPP/Util/SkillTracker.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/StateSetters/KickoffState.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>
#include <torch/torch.h>
#include <algorithm>
#include <cstring>
#include <filesystem>
#include <map>

using namespace RLGPC;
using namespace RLGSC;

namespace {
// every episode starts as a kickoff with the ball already behind a goal line: goal_sign +1 = the orange goal (blue scores), -1 = the blue goal
struct BallInGoalState : KickoffState {
    float goalSign;
    explicit BallInGoalState(float s) : goalSign(s) {}
    GameState ResetState(Arena* arena) override {
        GameState gs = KickoffState::ResetState(arena);
        BallState bs; bs.pos = Vec(0, goalSign * 5400.f, 200.f); bs.vel = Vec(0, 0, 0);
        arena->ball->SetState(bs);
        return GameState(arena);
    }
};
float g_goalSign = 0;   // 0: plain kickoffs
EnvCreateResult MakeEnv() {
    std::vector<TerminalCondition*> terminal = {new NoTouchCondition(100000), new GoalScoreCondition()};
    StateSetter* setter = g_goalSign != 0 ? (StateSetter*)new BallInGoalState(g_goalSign) : new KickoffState();
    Match* match = new Match(new VelocityReward(), terminal, new DefaultOBS(), new DiscreteAction(), setter, 1, true);
    return {match, new Gym(match, 8)};
}
}  // namespace

extern "C" {

// what RocketSim::Init does with the files of <dir>/soccar, in name order (see oracle/ref_driver.cpp ref_init_dir)
int refs_init_dir(const char* dir) {
    if (RocketSim::GetStage() == RocketSim::RocketSimStage::INITIALIZED) return 0;
    std::vector<std::filesystem::path> files;
    const std::filesystem::path folder = std::filesystem::path(dir) / "soccar";
    if (!std::filesystem::exists(folder)) return -1;
    for (auto& entry : std::filesystem::directory_iterator(folder)) if (entry.path().extension() == ".cmf") files.push_back(entry.path());
    std::sort(files.begin(), files.end());
    std::map<GameMode, std::vector<RocketSim::FileData>> m;
    for (auto& f : files) { DataStreamIn in(f, false); m[GameMode::SOCCAR].push_back(in.data); }
    try { RocketSim::InitFromMem(m, true); } catch (std::exception&) { return -1; }
    return 0;
}

// SkillTracker::UpdateRatings (SkillTracker.cpp:72-86) over a script: n_sets rating sets under the mode "1v1"; step i takes winner[i] / loser[i]
// (indices of sets) and flags[i] (bit 0 updateWinner, bit 1 updateLoser); ratings_io holds the sets' ratings before and after; trace_out
// (n x n_sets) the ratings after every step.
int refs_elo_script(int n_sets, float* ratings_io, int n, const int* winner, const int* loser, const int* flags, float rating_inc, float* trace_out) {
    SkillTrackerConfig cfg; cfg.envCreateFunc = MakeEnv; cfg.numEnvs = 1; cfg.numThreads = 1; cfg.ratingInc = rating_inc;
    g_goalSign = 0;
    SkillTracker st(cfg);
    std::vector<SkillTracker::RatingSet> sets(n_sets);
    for (int i = 0; i < n_sets; i++) sets[i].data["1v1"] = ratings_io[i];
    for (int i = 0; i < n; i++) {
        st.UpdateRatings(sets[winner[i]], sets[loser[i]], flags[i] & 1, flags[i] & 2, "1v1");
        for (int k = 0; k < n_sets; k++) trace_out[(size_t)i * n_sets + k] = sets[k].data["1v1"];
    }
    for (int i = 0; i < n_sets; i++) ratings_io[i] = sets[i].data["1v1"];
    return 0;
}

// SkillTracker::RunGames (SkillTracker.cpp:152-257) over a script of n_calls calls with timesteps_delta[i].  goal_sign as above (0: kickoffs,
// nobody scores within the short sim time).  One env, one thread, one env step per evaluating call (sim_time is whatever gives one step).
// Before every call: the env's teamSwap and oldPolicyIndex.  After every call: runCounter, number of stored versions, timestepsSinceVersionMade,
// the current rating and the ratings of the stored versions (up to max_versions of them, the rest of the row is left 0).
// rows_out: n_calls x (6 + max_versions) floats = {teamSwap, oldPolicyIndex, runCounter, nVersions, timestepsSinceVersionMade, curRating, old...}.
int refs_run_script(float goal_sign, int n_calls, const int64_t* timesteps_delta, int update_interval, int64_t timesteps_per_version, int max_versions,
                    int start_with_version, float rating_inc, float sim_time, float* rows_out) {
    torch::manual_seed(1);
    g_goalSign = goal_sign;
    SkillTrackerConfig cfg; cfg.envCreateFunc = MakeEnv; cfg.numEnvs = 1; cfg.numThreads = 1; cfg.ratingInc = rating_inc; cfg.simTime = sim_time;
    cfg.updateInterval = update_interval; cfg.timestepsPerVersion = timesteps_per_version; cfg.maxVersions = max_versions;
    cfg.startWithVersion = start_with_version != 0; cfg.kickoffStatesOnly = false; cfg.perModeRatings = true;
    SkillTracker st(cfg);
    DiscretePolicy policy(89, 90, {32, 32}, torch::kCPU);
    const int W = 6 + max_versions;
    for (int i = 0; i < n_calls; i++) {
        float* row = rows_out + (size_t)i * W;
        std::memset(row, 0, sizeof(float) * W);
        row[0] = st.games[0].teamSwap ? 1.f : 0.f; row[1] = (float)st.games[0].oldPolicyIndex;
        st.RunGames(&policy, timesteps_delta[i]);
        row[2] = (float)st.runCounter; row[3] = (float)st.oldPolicies.size(); row[4] = (float)st.timestepsSinceVersionMade; row[5] = st.curRating.data["1v1"];
        if (st.oldPolicies.size() != st.oldRatings.size()) return -2;
        for (size_t k = 0; k < st.oldRatings.size() && (int)k < max_versions; k++) row[6 + k] = st.oldRatings[k].data["1v1"];
    }
    return 0;
}

}  // extern "C"
