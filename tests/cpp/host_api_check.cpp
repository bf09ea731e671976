// CPU-only checks of the C++ host layer (include/RLGymSim_CPP, include/RLGymPPO_CPP): the plugin classes translate to the
// device gym configuration exactly as rlgpu_default_gym_config describes the reference's example stack, unsupported plugins
// fail loudly with the reference's "RG FATAL ERROR" exception, and the small utility types behave like the reference's.
#include <RLGymPPO_CPP/Learner.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CommonRewards.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/CombinedReward.h>
#include <RLGymSim_CPP/Utils/RewardFunctions/ZeroSumReward.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/NoTouchCondition.h>
#include <RLGymSim_CPP/Utils/TerminalConditions/GoalScoreCondition.h>
#include <RLGymSim_CPP/Utils/OBSBuilders/DefaultOBS.h>
#include <RLGymSim_CPP/Utils/StateSetters/RandomState.h>
#include <RLGymSim_CPP/Utils/StateSetters/KickoffState.h>
#include <RLGymSim_CPP/Utils/ActionParsers/DiscreteAction.h>
#include <cstdio>
#include <cstring>
using namespace RLGSC; using namespace RLGPC;

#define CHECK(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while (0)

struct MyReward : RewardFunction {   // a user reward without a device form
    float GetReward(const PlayerData& p, const GameState&, const Action&) override { return p.phys.vel.Length(); }
};

int main() {
    // the example stack == the C-ABI's default config, field by field
    CombinedReward rew({{new FaceBallReward(), 0.1f}, {new VelocityPlayerToBallReward(), 0.5f}, {new VelocityBallToGoalReward(), 1.0f},
                        {new EventReward({.teamGoal = 1.f, .concede = -1.f}), 50.f}}, true);
    NoTouchCondition nt(150); GoalScoreCondition gs; DefaultOBS obs; DiscreteAction act; RandomState rs(true, true, true);
    Match match(&rew, {&nt, &gs}, &obs, &act, &rs, 1, true);
    RlgpuGymConfig got = match.ToDeviceConfig(8), want; rlgpu_default_gym_config(&want);
    CHECK(got.tick_skip == want.tick_skip && got.n_terms == want.n_terms && got.n_conds == want.n_conds);
    for (int i = 0; i < want.n_terms; i++) CHECK(got.terms[i].kind == want.terms[i].kind && got.terms[i].weight == want.terms[i].weight && got.terms[i].p0 == want.terms[i].p0);
    CHECK(std::memcmp(got.event_weights, want.event_weights, sizeof(want.event_weights)) == 0);
    CHECK(got.conds[0] == want.conds[0] && got.conds[1] == want.conds[1] && got.no_touch_max_steps == want.no_touch_max_steps);
    CHECK(got.setter_kind == want.setter_kind && got.rand_ball_speed == 1 && got.rand_car_speed == 1 && got.cars_on_ground == 1);
    CHECK(std::memcmp(got.pos_coef, want.pos_coef, 12) == 0 && got.vel_coef == want.vel_coef && got.ang_vel_coef == want.ang_vel_coef);
    CHECK(got.n_actions == 90 && act.GetActionAmount() == 90 && match.playerAmount == 2);
    // DiscreteAction: row 0 of the table and the parse path
    ActionSet parsed = act.ParseActions({0, 89}, GameState());
    CHECK(parsed.size() == 2 && parsed[0].throttle == -1.f && parsed[0].steer == -1.f);
    // zero-sum wrapper, kickoff setter, weights multiply through nested CombinedRewards
    ZeroSumReward zs(new CombinedReward({{new TouchBallReward(0.5f), 2.f}, {new SaveBoostReward(), 1.f}}, true), 0.3f, 0.9f);
    KickoffState ks;
    Match m2(&zs, {&gs}, &obs, &act, &ks, 2, true);
    RlgpuGymConfig c2 = m2.ToDeviceConfig(4);
    CHECK(c2.zero_sum == 1 && c2.team_spirit == 0.3f && c2.opp_scale == 0.9f && c2.n_terms == 2 && c2.terms[0].kind == RLGPU_RW_TOUCH_BALL && c2.terms[0].weight == 2.f && c2.terms[0].p0 == 0.5f);
    CHECK(c2.setter_kind == RLGPU_SS_KICKOFF && c2.n_conds == 1 && c2.tick_skip == 4 && m2.playerAmount == 4);
    // a reward with no device form is refused loudly, with the reference's error prefix
    MyReward mine; Match m3(&mine, {&gs}, &obs, &act, &rs);
    bool threw = false;
    try { m3.ToDeviceConfig(8); } catch (const std::runtime_error& e) { threw = std::string(e.what()).rfind("RG FATAL ERROR", 0) == 0; }
    CHECK(threw);
    // Report / AvgTracker / WelfordRunningStat
    Report r; r.AccumAvg("x", 2); r.AccumAvg("x", 4); r["n"] = 1234567;
    CHECK(r.GetAvg("x") == 3 && r.Has("n") && !r.Has("y") && r.SingleToString("n", true) == "n: 1,234,567");
    AvgTracker a; CHECK(std::isnan(a.Get())); a += 1.f; a += NAN; a += 3.f; CHECK(a.Get() == 2.f && a.count == 2);
    WelfordRunningStat w; CHECK(w.GetSTD() == 1.0); w.Increment({1.f, 2.f, 3.f, 4.f}, 4);
    CHECK(std::fabs(w.GetSTD() - std::sqrt(5.0 / 3.0)) < 1e-12 && w.count == 4);
    // PhysObj::Invert mirrors x and y
    PhysObj p; p.pos = Vec(1, 2, 3); p.rotMat.forward = Vec(0, 1, 0);
    PhysObj q = p.Invert(); CHECK(q.pos.x == -1 && q.pos.y == -2 && q.pos.z == 3 && q.rotMat.forward.y == -1);
    // config defaults of the reference
    LearnerConfig lc; CHECK(lc.numThreads == 8 && lc.numGamesPerThread == 16 && lc.ppo.epochs == 10 && lc.ppo.batchSize == 50000 && lc.gaeGamma == 0.99f && lc.maxReturnsPerStatsInc == 150);
    // MetricSender: JSON lines under $RLGPU_METRICS_DIR/<project>/<run id>.jsonl; a given run id continues its file
    {
        MetricSender ms("proj", "grp", "run \"7\"");
        CHECK(ms.curRunID.size() == 8);
        Report rep; rep["Policy Entropy"] = 4.25; rep["Cumulative Timesteps"] = 1234567; rep["bad"] = NAN;
        ms.Send(rep);
        MetricSender again("proj", "grp", "run \"7\"", ms.curRunID);
        CHECK(again.curRunID == ms.curRunID && again.filePath == ms.filePath);
        again.Send(rep);
        std::ifstream f(ms.filePath); std::string line; std::vector<std::string> lines;
        while (std::getline(f, line)) lines.push_back(line);
        CHECK(lines.size() == 3 && lines[0].find("\"_run\"") != std::string::npos && lines[0].find("run \\\"7\\\"") != std::string::npos);
        CHECK(lines[1] == lines[2] && lines[1].find("\"Policy Entropy\": 4.25") != std::string::npos && lines[1].find("\"bad\": null") != std::string::npos);
        std::printf("metrics file: %s\n", ms.filePath.string().c_str());
    }
    // RenderSender: the RocketSimVis datagram arrives on a local UDP socket exactly as ToJSON builds it
    {
        int rx = socket(AF_INET, SOCK_DGRAM, 0);
        sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = 0; inet_pton(AF_INET, "127.0.0.1", &a.sin_addr);
        CHECK(bind(rx, (sockaddr*)&a, sizeof a) == 0);
        socklen_t al = sizeof a; CHECK(getsockname(rx, (sockaddr*)&a, &al) == 0);
        timeval tv{2, 0}; setsockopt(rx, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        RenderSender rs("127.0.0.1", ntohs(a.sin_port));
        // (the GameState of tests/golden/make_sender_golden.py: its datagram, produced by the reference's own render_receiver.py, is compared in Python)
        GameState gs; gs.players.resize(2); gs.players[0].carId = 1; gs.players[1].carId = 2; gs.players[1].team = Team::ORANGE;
        for (PhysObj* q : {&gs.ball, &gs.players[0].phys, &gs.players[1].phys}) { q->pos = q->vel = q->angVel = Vec(0, 0, 0); q->rotMat.forward = Vec(1, 0, 0); q->rotMat.right = Vec(0, 1, 0); q->rotMat.up = Vec(0, 0, 1); }
        gs.ball.pos = Vec(1, 2, 93.15f); gs.ball.vel = Vec(-2300.f, 1e16f, 123456789.f); gs.ball.angVel = Vec(0, 0, 6);
        PlayerData& p1 = gs.players[1];
        p1.phys.pos = Vec(-100, 250.5f, 17); p1.phys.rotMat.forward = Vec(0.6f, -0.8f, 0); p1.phys.rotMat.right = Vec(0.8f, 0.6f, 0);
        p1.phys.vel = Vec(1234.5678f, -0.0001f, 1e-5f); p1.phys.angVel = Vec(0.1f, -5.5f, 2.25f);
        p1.boostPickups = 3; p1.carState.isDemoed = true; p1.carState.isOnGround = true; p1.ballTouchedStep = true; p1.hasFlip = true; p1.boostFraction = 0.33f;
        gs.players[0].boostFraction = 0.f; gs.players[0].hasFlip = false; gs.players[0].carState.isOnGround = false;
        gs.boostPads[3] = true;
        rs.Send(gs, ActionSet(2));
        char buf[8192]; ssize_t n = recv(rx, buf, sizeof buf, 0);
        CHECK(n > 0);
        std::string got(buf, (size_t)n), want = RenderSender::ToJSON(gs, ActionSet(2));
        CHECK(got == want && rs.sent == 1);
        CHECK(got.find("\"gamemode\": \"soccar\"") == 1 && got.find("\"pos\": [1.0, 2.0, 93.1500015258789]") != std::string::npos);
        CHECK(got.find("\"team_num\": 1") != std::string::npos && got.find("\"boost_amount\": 0.33000001311302185") != std::string::npos);
        CHECK(RenderSender::PyFloat(1e-5f) == "9.999999747378752e-06" && RenderSender::PyFloat(-0.0001f) == "-9.999999747378752e-05" && RenderSender::PyFloat(1e16f) == "1.0000000272564224e+16"
              && RenderSender::PyFloat(123456789.f) == "123456792.0" && RenderSender::PyFloat(0.5f) == "0.5" && RenderSender::PyFloat(-0.f) == "-0.0" && RenderSender::PyFloat(1e22f) == "9.999999778196308e+21");
        std::printf("render datagram: %s\n", got.c_str());
        close(rx);
    }
    // ---- the open plugin boundary, host side only (no GPU needed): which kinds go where, and the Arena facade ----
    {
        struct MyOBS : DefaultOBS { void AddPlayerToOBS(FList& o, const PlayerData& p, bool inv) override { DefaultOBS::AddPlayerToOBS(o, p, inv); o += 1.f; } };
        struct MySetter : StateSetter { GameState ResetState(Arena* a) override { a->ResetToRandomKickoff(7); return GameState(a); } };
        struct MyTerminal : TerminalCondition { bool IsTerminal(const GameState&) override { return false; } };
        MyOBS myObs; MySetter mySetter; MyTerminal myTerm;
        Match::DevicePlan all = match.PlanDevice(8);
        CHECK(!all.AnyHost() && all.cfg.one_team == 0);
        Match::DevicePlan p1 = Match(&rew, {&nt, &gs}, &myObs, &act, &rs, 1, true).PlanDevice(8);
        CHECK(p1.hostObs && !p1.hostReward && !p1.hostTerminal && !p1.hostSetter && !p1.hostParser);   // a subclass of a built-in does not get the built-in's device form
        Match::DevicePlan p2 = Match(&mine, {&nt, &myTerm}, &obs, &act, &mySetter, 2, false).PlanDevice(4);
        CHECK(p2.hostReward && p2.hostTerminal && p2.hostSetter && !p2.hostObs && p2.cfg.n_conds == 0 && p2.cfg.n_terms == 0 && p2.cfg.one_team == 1 && p2.cfg.tick_skip == 4);
        // Arena facade: slots, ids, GetState / SetState, kickoff with a seed, GameState(Arena*)
        Arena* arena = Arena::Create(GameMode::SOCCAR);
        Car* b0 = arena->AddCar(Team::BLUE); Car* o0 = arena->AddCar(Team::ORANGE); Car* b1 = arena->AddCar(Team::BLUE); Car* o1 = arena->AddCar(Team::ORANGE);
        CHECK(b0->id == 1 && o0->id == 2 && b1->id == 3 && o1->id == 4 && arena->_state.num_cars == 4 && arena->GetCar(3) == b1 && arena->_boostPads.size() == 34);
        bool threw2 = false;
        try { arena->AddCar(Team::ORANGE); } catch (const std::runtime_error&) { threw2 = true; }   // blue is next
        CHECK(threw2);
        CarState cs; cs.pos = Vec(100, -200, 300); cs.vel = Vec(1, 2, 3); cs.boost = 55; cs.isOnGround = false; cs.hasFlipped = true; cs.worldContact.hasContact = true;
        cs.ballHitInfo.isValid = true; cs.ballHitInfo.tickCountWhenHit = 77; cs.lastControls.jump = true; cs.rotMat = Angle(0.3f, 0.1f, -0.2f).ToRotMat();
        b1->SetState(cs);
        CarState back = b1->GetState();
        CHECK(back.pos.x == 100 && back.vel.z == 3 && back.boost == 55 && !back.isOnGround && back.hasFlipped && back.worldContact.hasContact && back.ballHitInfo.isValid &&
              back.ballHitInfo.tickCountWhenHit == 77 && back.lastControls.jump && back.rotMat.forward.x == cs.rotMat.forward.x && back.rotMat.up.z == cs.rotMat.up.z);
        const float fx = cs.rotMat.forward.x, fy = cs.rotMat.forward.y, fz = cs.rotMat.forward.z;
        CHECK(std::fabs(fx * fx + fy * fy + fz * fz - 1.f) < 1e-6f && std::fabs(cs.rotMat.forward.Dot(cs.rotMat.up)) < 1e-6f);   // Angle::ToRotMat is a rotation
        arena->ResetToRandomKickoff(5);
        const CarState kb = b0->GetState(), ko = o0->GetState();
        CHECK(kb.pos.y < 0 && kb.pos.z == 17.f && ko.pos.x == -kb.pos.x && ko.pos.y == -kb.pos.y && arena->ball->GetState().pos.z == RLConst::BALL_REST_Z);
        Arena* again = Arena::Create(GameMode::SOCCAR);
        again->AddCar(Team::BLUE); again->AddCar(Team::ORANGE); again->ResetToRandomKickoff(5);
        CHECK(again->_cars[0]->GetState().pos.x == kb.pos.x && again->_cars[0]->GetState().pos.y == kb.pos.y);   // same seed, same spot
        GameState fresh(arena);
        CHECK(fresh.players.size() == 4 && fresh.players[2].carId == 3 && fresh.players[1].team == Team::ORANGE && fresh.players[0].phys.pos.y == kb.pos.y);
        // the reference's RandomState on the facade: everything inside its ranges, cars on the ground flat
        RocketSim::Math::SeedRandEngine(3);
        RandomState scatter(true, true, true);
        for (int i = 0; i < 50; i++) {
            GameState st = scatter.ResetState(arena);
            CHECK(std::fabs(st.ball.pos.x) <= 3500 && std::fabs(st.ball.pos.y) <= 4000 && st.ball.pos.z >= 92.75f && st.ball.pos.z <= 1820 && st.ball.vel.Length() <= 4000.01f);
            for (const PlayerData& pd : st.players) CHECK(pd.phys.pos.z == 17.f && pd.phys.vel.z == 0 && pd.phys.rotMat.up.z > 0.9999f && pd.carState.boost >= 0 && pd.carState.boost <= 100);
        }
        // one-team arena: blue cars only, ids 1, 2, 3; the orange slots stay empty
        Arena* solo = Arena::Create(GameMode::SOCCAR);
        solo->AddCar(Team::BLUE); solo->AddCar(Team::BLUE); solo->AddCar(Team::BLUE);
        CHECK(solo->_IsOneTeam() && solo->_state.num_cars == 6 && solo->_cars[2]->id == 3 && (solo->_state.cars[1].flags & RLGPU_CF_ABSENT) && !(solo->_state.cars[2].flags & RLGPU_CF_ABSENT));
        GameState soloState(solo);
        CHECK(soloState.players.size() == 3 && soloState.players[2].carId == 3);
        delete arena; delete again; delete solo;
    }
    std::printf("host api ok\n");
    return 0;
}
