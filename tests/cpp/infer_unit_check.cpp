// GPU check of the deployment path:
//  1. the host OBS builders (DefaultOBS / DefaultOBSPadded::BuildOBS on a GameState materialised from the device state) give
//     bit-for-bit the rows the device builder wrote for the same step (tickSkip 1, so the observed snapshot IS the final state);
//  2. InferUnit loads a PPO_POLICY.lt / PPO_CRITIC.lt archive and infers on those states; obs rows, distributions, deterministic
//     actions and values go to a file the Python test compares with the numpy oracle.
// usage: infer_unit_check <policy.lt> <critic.lt> <out.bin>
#include <hip/hip_runtime.h>
#include <RLGymPPO_CPP/Util/InferUnit.h>
#include <RLGymPPO_CPP/Util/SkillTracker.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/StateSetters/RandomState.h>
#include <RLGymSim_CPP/Utils/StateSetters/KickoffState.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBSPadded.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>
#include "../../include/rlgpu_state.h"
#include <cstdio>
#include <cstring>
#include <fstream>
using namespace RLGSC; using namespace RLGPC;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)
#define HIPOK(call) CHECK((call) == hipSuccess)

static int obs_parity(int teamSize, int nEnvs, int steps, std::vector<GameState>* keepStates, std::vector<ActionSet>* keepPrev) {
    RlgpuGymConfig g; rlgpu_default_gym_config(&g);
    g.tick_skip = 1; g.seed_lo = 77 + teamSize; g.no_touch_max_steps = 40;
    rlgpu_env* env = nullptr;
    CHECK(rlgpu_env_create(&env, 0, nEnvs, teamSize, &g) == RLGPU_OK);
    CHECK(rlgpu_env_set_procedural_mesh(env) == RLGPU_OK);
    const int nAgents = rlgpu_env_num_agents(env), D = rlgpu_env_obs_size(env), P = 2 * teamSize;
    float *obs = nullptr, *rew = nullptr; int32_t *acts = nullptr, *done = nullptr;
    HIPOK(hipMalloc(&obs, (size_t)nAgents * D * 4)); HIPOK(hipMalloc(&rew, nAgents * 4)); HIPOK(hipMalloc(&acts, nAgents * 4)); HIPOK(hipMalloc(&done, nAgents * 4));
    CHECK(rlgpu_env_reset(env, 1, obs) == RLGPU_OK);
    CHECK(rlgpu_env_enable_snapshots(env, 1) == RLGPU_OK);   // every step's GameState source = the arena where the episode ended, for the steps that end one
    DefaultOBS builder; DiscreteAction parser;
    std::vector<int32_t> hostActs(nAgents), hostDone(nAgents);
    std::vector<float> hostObs((size_t)nAgents * D);
    std::vector<RlgpuArenaState> states(nEnvs), snaps(nEnvs);
    uint32_t rng = 12345;
    int compared = 0, resets = 0;
    for (int s = 0; s < steps; s++) {
        for (int& a : hostActs) { rng = rng * 1664525u + 1013904223u; a = (int)((rng >> 8) % 90u); }
        HIPOK(hipMemcpy(acts, hostActs.data(), nAgents * 4, hipMemcpyHostToDevice));
        CHECK(rlgpu_env_step(env, acts, obs, rew, done) == RLGPU_OK);
        CHECK(rlgpu_env_sync(env) == RLGPU_OK);
        HIPOK(hipMemcpy(hostObs.data(), obs, hostObs.size() * 4, hipMemcpyDeviceToHost));
        HIPOK(hipMemcpy(hostDone.data(), done, nAgents * 4, hipMemcpyDeviceToHost));
        CHECK(rlgpu_env_download_states(env, states.data(), nullptr, nEnvs) == RLGPU_OK);
        CHECK(rlgpu_env_download_snapshots(env, snaps.data(), 0, nEnvs) == RLGPU_OK);
        for (int e = 0; e < nEnvs; e++) {
            // a new episode's first GameState is the one its state setter built.  This env's setter is the built-in RandomState, which resets the
            // pads before it builds it (RandomState.cpp:11 -> Arena::ResetToRandomKickoff, Arena.cpp:209-210): the downloaded state (all pads
            // active after the reset) is what the first observation shows.  (Only a user setter that leaves the pads alone shows the previous
            // episode's: Match.cpp:55-69, tests/golden/gameinst_golden.npz.)
            GameState gs(states[e], 1);
            CHECK((int)gs.players.size() == P);
            IList idx(hostActs.begin() + e * P, hostActs.begin() + (e + 1) * P);
            ActionSet prev = parser.ParseActions(idx, gs);
            if (hostDone[e * P]) {   // the recorded row is the first observation of the new episode: previous actions are cleared (Match.cpp:28-30)
                resets++;
                for (Action& a : prev) a = Action();
            }
            for (int k = 0; k < P; k++) {
                FList row = builder.BuildOBS(gs.players[k], gs, prev[k]);
                CHECK((int)row.size() == D);
                const float* dev = &hostObs[(size_t)(e * P + k) * D];
                for (int i = 0; i < D; i++)
                    if (std::memcmp(&row[i], &dev[i], 4) != 0) {
                        std::printf("obs mismatch: team size %d step %d env %d player %d column %d host %.9g device %.9g\n", teamSize, s, e, k, i, row[i], dev[i]);
                        return 1;
                    }
                compared++;
            }
            if (keepStates && e == 0 && (s % 7) == 3) { keepStates->push_back(gs); keepPrev->push_back(prev); }
        }
    }
    std::printf("obs parity: team size %d, %d rows bit-exact (%d episode resets seen)\n", teamSize, compared, resets);
    CHECK(resets > 0);
    // the padded builder at maxPlayers == team size only reorders the other players' blocks
    {
        DefaultOBSPadded padded(teamSize);
        GameState gs(states[0], 1);
        FList a = builder.BuildOBS(gs.players[0], gs, Action()), b = padded.BuildOBS(gs.players[0], gs, Action());
        CHECK(a.size() == b.size());
        CHECK(std::memcmp(a.data(), b.data(), 70 * 4) == 0);
        auto blocks = [&](const FList& r, int from, int n) { std::vector<std::vector<float>> v; for (int i = 0; i < n; i++) v.push_back(std::vector<float>(r.begin() + from + 19 * i, r.begin() + from + 19 * (i + 1))); std::sort(v.begin(), v.end()); return v; };
        CHECK(blocks(a, 70, teamSize - 1) == blocks(b, 70, teamSize - 1));
        CHECK(blocks(a, 70 + 19 * (teamSize - 1), teamSize) == blocks(b, 70 + 19 * (teamSize - 1), teamSize));
        // wider padding adds zero blocks
        DefaultOBSPadded wide(teamSize + 1);
        FList w = wide.BuildOBS(gs.players[0], gs, Action());
        CHECK((int)w.size() == 70 + 19 * (2 * (teamSize + 1) - 1));
        int zeroBlocks = 0;
        for (auto& blk : blocks(w, 70, 2 * (teamSize + 1) - 1)) zeroBlocks += std::all_of(blk.begin(), blk.end(), [](float x) { return x == 0; });
        CHECK(zeroBlocks == 2);
    }
    (void)hipFree(obs); (void)hipFree(rew); (void)hipFree(acts); (void)hipFree(done);
    rlgpu_env_destroy(env);
    return 0;
}

static EnvCreateResult EvalEnv() {
    Match* match = new Match(new VelocityReward(), {new NoTouchCondition(40), new GoalScoreCondition()}, new DefaultOBS(), new DiscreteAction(), new RandomState(true, true, true), 1, true);
    return {match, new Gym(match, 8)};
}

// SkillTracker: rating arithmetic, the rating (de)serialisation, version bookkeeping, and eval games on the device
static int skill_tracker_check() {
    RlgpuLearnerConfig lc{};
    lc.obs_size = 89; lc.n_actions = 90; lc.n_policy_layers = 2; lc.n_critic_layers = 1; lc.policy_layers[0] = 64; lc.policy_layers[1] = 64; lc.critic_layers[0] = 16;
    lc.clip_range = 0.2f; lc.temperature = 1; lc.seed_lo = 5; lc.max_rows = 256;
    rlgpu_learner* lrn = nullptr;
    CHECK(rlgpu_learner_create(&lrn, 0, &lc) == RLGPU_OK);
    SkillTrackerConfig sc;
    sc.enabled = true; sc.envCreateFunc = EvalEnv; sc.numEnvs = 3; sc.simTime = 12; sc.updateInterval = 2; sc.timestepsPerVersion = 1000; sc.maxVersions = 2;
    int callbacks = 0;
    sc.stepCallback = [&](GameInst* g, const Gym::StepResult& r, Report&) { callbacks += g->isEval && r.state.players.size() == 2 && r.reward.size() == 2; };
    SkillTracker st(sc, lrn, 89, 90, {64, 64}, 123);
    CHECK(st.modeName == "1v1" && st.curRating.data.at("1v1") == 1000.f && st.games.size() == 3);
    // elo step (SkillTracker.cpp:72-86): equal ratings -> +-ratingInc/2; a 400 point favourite gains 1/11 of the increment
    SkillTracker::RatingSet a, b; a.data["1v1"] = 1000; b.data["1v1"] = 1000;
    st.UpdateRatings(a, b, true, true, "1v1");
    CHECK(a.data["1v1"] == 1002.5f && b.data["1v1"] == 997.5f);
    a.data["1v1"] = 1400; b.data["1v1"] = 1000;
    st.UpdateRatings(a, b, true, false, "1v1");
    CHECK(std::fabs(a.data["1v1"] - (1400 + 5.f / 11)) < 1e-3f && b.data["1v1"] == 1000);
    // rating sets in RUNNING_STATS.json: per-mode object, bare number, missing mode (SkillTracker.cpp:259-291)
    CHECK(st.LoadRatingSet("{\"1v1\": 1234.5, \"2v2\": 900}").data.at("1v1") == 1234.5f);
    CHECK(st.LoadRatingSet("  1100.25\n}").data.at("1v1") == 1100.25f);
    CHECK(st.LoadRatingSet("{\"3v3\": 1}", false).data.at("1v1") == 1000.f);
    CHECK(st.RatingsToJSON() == "{\"1v1\": 1000}");
    // run 0 evaluates (and stores the first version), run 1 is skipped by updateInterval, versions are capped at maxVersions
    st.RunGames(600);
    CHECK(st.NumOldPolicies() == 1 && st.oldRatings.size() == 1 && st.runCounter == 1);
    const int stepsPerRun = (int)(12.f / 3 * 120 / 8);
    CHECK(callbacks == 3 * stepsPerRun);
    st.RunGames(600);
    CHECK(callbacks == 3 * stepsPerRun && st.NumOldPolicies() == 1 && st.timestepsSinceVersionMade == 600);   // skipped entirely, like the reference's early return
    st.RunGames(600);
    CHECK(callbacks == 6 * stepsPerRun && st.NumOldPolicies() == 2 && st.timestepsSinceVersionMade == 0);
    for (int i = 0; i < 4; i++) st.RunGames(1000);
    CHECK(st.NumOldPolicies() == 2 && st.oldRatings.size() == 2);
    for (auto& g : st.games) CHECK(g.oldPolicyIndex >= 0 && g.oldPolicyIndex < 2);
    std::printf("skill tracker ok: rating %.3f after %d eval steps\n", st.curRating.data.at("1v1"), callbacks);
    rlgpu_learner_destroy(lrn);
    return 0;
}

// The recordings of the real reference skill tracker (tests/golden/skill_golden.json, flattened to text by tests/test_host_cpp.py):
//   ELO  the ratings after every scripted UpdateRatings step, bit for bit
//   RUN  scripted RunGames calls: the version bookkeeping (runCounter, stored versions, timestepsSinceVersionMade) equals the recording's;
//        with every episode starting behind a goal line each evaluating call is one goal, and the ratings move as the reference's rule says
//        for THIS tracker's teamSwap / oldPolicyIndex (drawn from a wall-clock seeded engine there, so the recording's own draws differ)
static float g_goalSign = 0;
struct BallInGoalState : KickoffState {
    GameState ResetState(Arena* arena) override {
        GameState gs = KickoffState::ResetState(arena);
        BallState bs; bs.pos = Vec(0, g_goalSign * 5400.f, 200.f); bs.vel = Vec(0, 0, 0);
        arena->ball->SetState(bs);
        return GameState(arena);
    }
};
static EnvCreateResult ScriptEnv() {
    StateSetter* setter = g_goalSign != 0 ? (StateSetter*)new BallInGoalState() : new KickoffState();
    Match* match = new Match(new VelocityReward(), {new NoTouchCondition(100000), new GoalScoreCondition()}, new DefaultOBS(), new DiscreteAction(), setter, 1, true);
    return {match, new Gym(match, 8)};
}
static float from_bits(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static int skill_fixture_check(const char* path) {
    FILE* f = std::fopen(path, "r");
    CHECK(f != nullptr);
    RlgpuLearnerConfig lc{};
    lc.obs_size = 89; lc.n_actions = 90; lc.n_policy_layers = 2; lc.n_critic_layers = 1; lc.policy_layers[0] = 32; lc.policy_layers[1] = 32; lc.critic_layers[0] = 16;
    lc.clip_range = 0.2f; lc.temperature = 1; lc.seed_lo = 5; lc.max_rows = 256;
    rlgpu_learner* lrn = nullptr;
    CHECK(rlgpu_learner_create(&lrn, 0, &lc) == RLGPU_OK);
    char tag[16]; int nElo = 0, nRuns = 0, nGoals = 0;
    while (std::fscanf(f, "%15s", tag) == 1) {
        if (std::string(tag) == "ELO") {
            int nSets, n; float inc;
            CHECK(std::fscanf(f, "%d %d %f", &nSets, &n, &inc) == 3);
            g_goalSign = 0;
            SkillTrackerConfig sc; sc.enabled = true; sc.envCreateFunc = ScriptEnv; sc.numEnvs = 1; sc.ratingInc = inc;
            SkillTracker st(sc, lrn, 89, 90, {32, 32}, 1);
            std::vector<SkillTracker::RatingSet> sets(nSets);
            for (int k = 0; k < nSets; k++) { unsigned u; CHECK(std::fscanf(f, "%u", &u) == 1); sets[k].data["1v1"] = from_bits(u); }
            for (int i = 0; i < n; i++) {
                int w, l, fl; CHECK(std::fscanf(f, "%d %d %d", &w, &l, &fl) == 3);
                st.UpdateRatings(sets[w], sets[l], fl & 1, fl & 2, "1v1");
                for (int k = 0; k < nSets; k++) {
                    unsigned u; CHECK(std::fscanf(f, "%u", &u) == 1);
                    const float mine = sets[k].data["1v1"]; unsigned mu; std::memcpy(&mu, &mine, 4);
                    if (mu != u) { std::printf("elo step %d set %d: %.9g, the reference has %.9g\n", i, k, mine, from_bits(u)); return 1; }
                }
            }
            nElo += n;
        } else if (std::string(tag) == "RUN") {
            float goalSign, inc, simTime; int nCalls, interval, maxVersions, startWith; long long perVersion;
            CHECK(std::fscanf(f, "%f %d %d %lld %d %d %f %f", &goalSign, &nCalls, &interval, &perVersion, &maxVersions, &startWith, &inc, &simTime) == 8);
            g_goalSign = goalSign;
            SkillTrackerConfig sc; sc.enabled = true; sc.envCreateFunc = ScriptEnv; sc.numEnvs = 1; sc.ratingInc = inc; sc.simTime = simTime;
            sc.updateInterval = interval; sc.timestepsPerVersion = perVersion; sc.maxVersions = maxVersions; sc.startWithVersion = startWith != 0; sc.kickoffStatesOnly = false;
            SkillTracker st(sc, lrn, 89, 90, {32, 32}, 77 + nRuns);
            float cur = sc.initialRating; std::vector<float> olds;
            SkillTracker::RatingSet a, b;
            for (int i = 0; i < nCalls; i++) {
                long long delta; int refSwap, refIdx, refCounter, refVersions; long long refSince; unsigned curBits;
                CHECK(std::fscanf(f, "%lld %d %d %d %d %lld %u", &delta, &refSwap, &refIdx, &refCounter, &refVersions, &refSince, &curBits) == 7);
                for (int k = 0; k < maxVersions; k++) { unsigned u; CHECK(std::fscanf(f, "%u", &u) == 1); }
                const bool swap = st.games[0].teamSwap; int idx = st.games[0].oldPolicyIndex;
                const bool evaluates = st.runCounter % (uint64_t)interval == 0;
                // the rule (SkillTracker.cpp:104-146, 152-257) applied to this tracker's own draws
                if (evaluates) {
                    if (olds.empty() && startWith) olds.push_back(cur);
                    if (!olds.empty() && goalSign != 0) {
                        if (idx >= (int)olds.size()) idx = (int)olds.size() - 1;
                        const bool curScored = (goalSign > 0) != swap;
                        a.data["1v1"] = curScored ? cur : olds[idx]; b.data["1v1"] = curScored ? olds[idx] : cur;
                        st.UpdateRatings(a, b, true, true, "1v1");
                        (curScored ? cur : olds[idx]) = a.data["1v1"]; (curScored ? olds[idx] : cur) = b.data["1v1"];
                        nGoals++;
                    }
                }
                st.RunGames(delta);
                if (evaluates && st.timestepsSinceVersionMade == 0) { olds.push_back(cur); if ((int)olds.size() > maxVersions) olds.erase(olds.begin()); }
                if ((int)st.runCounter != refCounter || st.NumOldPolicies() != refVersions || st.timestepsSinceVersionMade != refSince || (int)st.oldRatings.size() != refVersions) {
                    std::printf("run script call %d: runCounter %d versions %d since %lld, the reference has %d %d %lld\n", i, (int)st.runCounter, st.NumOldPolicies(),
                                (long long)st.timestepsSinceVersionMade, refCounter, refVersions, refSince);
                    return 1;
                }
                CHECK(st.curRating.data.at("1v1") == cur && olds.size() == st.oldRatings.size());
                for (size_t k = 0; k < olds.size(); k++) CHECK(st.oldRatings[k].data.at("1v1") == olds[k]);
                if (goalSign == 0) CHECK(from_bits(curBits) == cur);   // nobody scores from a kickoff in one step, there or here
            }
            nRuns++;
        } else { std::printf("skill script: unknown tag %s\n", tag); return 1; }
    }
    std::fclose(f);
    CHECK(nElo >= 100 && nRuns >= 4 && nGoals >= 60);
    rlgpu_learner_destroy(lrn);
    std::printf("skill tracker vs the reference's recordings ok: %d elo steps bit-equal, %d scripted runs, %d goals rated\n", nElo, nRuns, nGoals);
    return 0;
}

int main(int argc, char** argv) {
    if (argc == 3 && std::string(argv[1]) == "--skill-script") return skill_fixture_check(argv[2]);
    if (argc < 4) return 2;
    if (skill_tracker_check()) return 1;
    std::vector<GameState> states; std::vector<ActionSet> prevs;
    if (obs_parity(1, 16, 120, &states, &prevs)) return 1;
    if (obs_parity(2, 8, 60, nullptr, nullptr)) return 1;
    if (obs_parity(3, 4, 60, nullptr, nullptr)) return 1;
    CHECK(states.size() >= 8);

    DefaultOBS obs; DiscreteAction parser;
    InferUnit policy(&obs, &parser, argv[1], true, 89, {32, 24});
    InferUnit critic(&obs, &parser, argv[2], false, 89, {16});
    std::ofstream out(argv[3], std::ios::binary);
    auto put = [&](const FList& v) { out.write((const char*)v.data(), (std::streamsize)v.size() * 4); };
    int32_t n = (int32_t)states.size();
    out.write((const char*)&n, 4);
    for (size_t i = 0; i < states.size(); i++) {
        const GameState& gs = states[i];
        const PlayerData& p = gs.players[i % 2];
        const Action& prev = prevs[i][i % 2];
        put(policy.GetObs(p, gs, prev));
        FList probs = policy.InferPolicySingleDistrib(p, gs, prev);
        CHECK(probs.size() == 90);
        put(probs);
        put(policy.InferPolicySingleDistrib(p, gs, prev, 2.5f));                  // temperature
        put(policy.InferPolicySingle(p, gs, prev, true).ToFList());              // deterministic: the arg-max row of the action table
        ActionSet all = policy.InferPolicyAll(gs, prevs[i], true);
        CHECK(all.size() == 2);
        CHECK(std::memcmp(&all[i % 2], &(const Action&)policy.InferPolicySingle(p, gs, prev, true), sizeof(Action)) == 0);
        Action sampled = policy.InferPolicySingle(p, gs, prev, false);           // sampled: some row of the table
        bool inTable = false;
        for (const Action& a : parser.actions) inTable = inTable || std::memcmp(&a, &sampled, sizeof(Action)) == 0;
        CHECK(inTable);
        FList vals = critic.InferCriticAll(gs, prevs[i]);
        CHECK(vals.size() == 2 && vals[i % 2] == critic.InferCriticSingle(p, gs, prev));
        put({vals[i % 2]});
    }
    // asking a policy unit for values (and the reverse) is the reference's error
    bool threw = false;
    try { policy.InferCriticSingle(states[0].players[0], states[0], Action()); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("created to infer the") != std::string::npos; }
    CHECK(threw);
    threw = false;
    try { InferUnit bad(&obs, &parser, argv[1], true, 89, {32, 32}); } catch (const std::runtime_error& e) { threw = std::string(e.what()).find("different model arch") != std::string::npos; }
    CHECK(threw);
    std::printf("infer unit ok\n");
    return 0;
}
