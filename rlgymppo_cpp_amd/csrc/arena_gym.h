// arena_gym.h — the gym layer around the tick, per env: RLGSC::Gym::Step / Reset (RLGymSim_CPP/src/RLGymSim_CPP/
// Gym.cpp:58-102), Match (Envs/Match.cpp:4-70), GameState snapshot (Utils/Gamestates/GameState.cpp:52-104,
// PlayerData.cpp:4-34), GameEventTracker::Update (RocketSim/src/Sim/GameEventTracker/GameEventTracker.cpp:48-158),
// and device fast paths of the built-in plugins: DiscreteAction, DefaultOBS, the CommonRewards stack,
// CombinedReward / ZeroSumReward, NoTouch / GoalScore conditions, RandomState / KickoffState.
#pragma once
#include "arena_step.h"
#include "arena_io.h"

namespace rlg {

enum RewardKind : int32_t {
    RW_EVENT = 0,              // EventReward (CommonRewards.cpp:4-42); weights in GymConfig::event_weights
    RW_VELOCITY = 1,           // VelocityReward(isNegative = p0 != 0)
    RW_SAVE_BOOST = 2,         // SaveBoostReward(exponent = p0)
    RW_VEL_BALL_TO_GOAL = 3,   // VelocityBallToGoalReward(ownGoal = p0 != 0)
    RW_VEL_PLAYER_TO_BALL = 4, // VelocityPlayerToBallReward
    RW_FACE_BALL = 5,          // FaceBallReward
    RW_TOUCH_BALL = 6          // TouchBallReward(aerialWeight = p0)
};
enum TerminalKind : int32_t { TC_NO_TOUCH = 0, TC_GOAL_SCORE = 1 };
enum SetterKind : int32_t { SS_RANDOM = 0, SS_KICKOFF = 1 };

struct RewardTerm { int32_t kind; float weight; float p0; };

struct GymConfig {
    int32_t tick_skip;
    int32_t n_terms; RewardTerm terms[8];          // CombinedReward order
    float event_weights[RLGPU_NUM_EVENT_VALS];
    int32_t zero_sum; float team_spirit, opp_scale; // ZeroSumReward wrapper (ZeroSumReward.cpp:3-29)
    int32_t n_conds; int32_t conds[4]; int32_t no_touch_max_steps;
    int32_t setter_kind; int32_t rand_ball_speed, rand_car_speed, cars_on_ground;
    uint32_t seed_lo, seed_hi;
    float pos_coef[3], vel_coef, ang_vel_coef;      // DefaultOBS.h:11-15
    int32_t n_actions;
    int32_t obs_max_players;   // 0 DefaultOBS, else DefaultOBSPadded(maxPlayers), team size <= maxPlayers <= 4: mates / opponents padded with zero blocks and shuffled
    int32_t one_team;          // Match(..., spawnOpponents = false): only the blue slots (even k) hold a car; see player_present()
    int32_t host_resets;       // 1: a step that ends an episode leaves the env as the episode left it; the host runs its state setter and resets the env (rlgpu_env_reset_envs)
};
// key of the tick's random draws (the respawn spot of a demolished car): both seed words, so that a resumed run (rlgpu_env_reseed bumps
// seed_hi, the epoch) does not replay the draws of the run it continues; seed_hi = 0 gives the key of the first epoch
RLG_HD uint32_t tick_seed(const GymConfig& cfg) { return cfg.seed_lo ^ 0xA511E9B3u ^ (cfg.seed_hi * 0x9E3779B9u); }

// spawnOpponents = false (Gym.cpp:45-49 adds no orange cars).  The env keeps its 2 * teamSize slots -- slot parity IS the team everywhere in
// the stepper -- and the orange ones are ABSENT: parked as demolished cars whose respawn timer never runs out (so the physics never sees
// them), skipped by the obs builder, the rewards and the agent rows.  Agent row of slot k inside its env: k / 2.
RLG_HD bool player_present(const GymConfig& cfg, int k) { return !cfg.one_team || (k % 2) == 0; }
RLG_HD int agent_row(const GymConfig& cfg, int k) { return cfg.one_team ? (k >> 1) : k; }
template <int NC>
RLG_HD int players_per_env(const GymConfig& cfg) { return cfg.one_team ? NC / 2 : NC; }
constexpr float ABSENT_RESPAWN_TIMER = 1e30f;

// obs-order -> RocketSim pad index (GameState.cpp:10-50 builds this by matching CommonValues::BOOST_LOCATIONS
// against RLConst pad positions; it is a constant of the two tables)
RLG_HD int pad_obs_to_rs(int i) {
    const int8_t M[34] = {6, 7, 8, 4, 5, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 0, 19, 20, 1, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 2, 3, 31, 32, 33};
    return M[i];
}

// DiscreteAction lookup table row (DiscreteAction.cpp:3-67): throttle steer pitch yaw roll jump boost handbrake
inline int build_action_table(float* out /*[90*8]*/) {
    const float RB[2] = {0, 1}, RF[3] = {-1, 0, 1};
    int n = 0;
    for (float throttle : RF) for (float steer : RF) for (float boost : RB) for (float handbrake : RB) {
        if (boost == 1 && throttle != 1) continue;
        float r[8] = {throttle, steer, 0, steer, 0, 0, boost, handbrake};
        for (int k = 0; k < 8; k++) out[n * 8 + k] = r[k];
        n++;
    }
    for (float pitch : RF) for (float yaw : RF) for (float roll : RF) for (float jump : RB) for (float boost : RB) {
        if (jump == 1 && yaw != 0) continue;
        if (pitch == roll && roll == jump && jump == 0) continue;
        float handbrake = (jump == 1) && (pitch != 0 || yaw != 0 || roll != 0);
        float r[8] = {boost, yaw, pitch, yaw, roll, jump, boost, handbrake};
        for (int k = 0; k < 8; k++) out[n * 8 + k] = r[k];
        n++;
    }
    return n;
}

// ---- snapshot (GameState::UpdateFromArena) ---------------------------------------------------------------
template <int NC>
struct Snapshot {
    V3 ball_pos, ball_vel, ball_angvel;                 // uu
    V3 car_pos[NC], car_fwd[NC], car_up[NC], car_vel[NC], car_angvel[NC];
    float boost_frac[NC];
    bool on_ground[NC], has_flip[NC], demoed[NC], touched[NC];
    uint64_t pads_active;                               // bit i = obs-order pad i active
    uint32_t car_order;                                 // Arena::car_order: GameState::players lists the cars in this order (the iteration order of Arena::_cars)
};

template <int NC>
RLG_HD_T6A void take_snapshot(const Arena<NC>& A, GymEnv<NC>& G, Snapshot<NC>& S) {
    int64_t tick_skip = A.tick_count - G.last_tick_count; if (tick_skip < 0) tick_skip = 0;
    S.ball_pos = A.ball.b.pos * BT2UU; S.ball_vel = A.ball.b.vel * BT2UU; S.ball_angvel = A.ball.b.angvel;
    for (int k = 0; k < NC; k++) {
        const Car& c = A.cars[k];
        S.car_pos[k] = c.b.pos * BT2UU; S.car_fwd[k] = col0(c.b.rot); S.car_up[k] = col2(c.b.rot);
        S.car_vel[k] = c.b.vel * BT2UU; S.car_angvel[k] = c.b.angvel;
        S.touched[k] = (c.flags & CF_BALLHIT_VALID) && (c.bh_tick_hit >= (A.tick_count - tick_skip));  // PlayerData.cpp:20-25
        if (S.touched[k]) G.last_touch_car_id = k + 1;
        S.has_flip[k] = !(c.flags & CF_HAS_DOUBLE_JUMPED) && !(c.flags & CF_HAS_FLIPPED) && c.air_time_since_jump < K::DOUBLEJUMP_MAX_DELAY;
        S.on_ground[k] = c.flags & CF_ON_GROUND; S.demoed[k] = c.flags & CF_IS_DEMOED;
        S.boost_frac[k] = c.boost / 100;
    }
    uint64_t m = 0;
    for (int i = 0; i < 34; i++) if (A.pads[pad_obs_to_rs(i)].is_active) m |= (1ull << i);
    S.pads_active = m;
    S.car_order = A.car_order;
    if (fabsf(S.ball_pos.y) > K::GOAL_THRESHOLD_Y + K::BALL_RADIUS) {  // GameState.cpp:100-101, Math.cpp:3-5
        int team_from_y = S.ball_pos.y < 0 ? 0 : 1;
        G.score_line[1 - team_from_y]++;
    }
    G.last_tick_count = A.tick_count;
}

// ---- step records (rlgpu_state.h RlgpuStepHead / RlgpuStepCar): a GameState's worth of an env, for host plugins that run after the launch -------------
// `touched_step`: PlayerData::ballTouchedStep per car as take_snapshot found it (the window ends where the previous GameState was taken, and
// take_snapshot has moved that mark already); null = a freshly reset arena (nobody has touched anything).
template <int NC>
RLG_HD void write_step_record(const Arena<NC>& A, const GymEnv<NC>& G, const bool* touched_step, bool done, uint32_t* o) {
    static_assert(sizeof(RlgpuStepHead) == 72 && sizeof(RlgpuStepCar) == 132, "step record layout");
    auto F = [](float f) { return f2u(f); };
    auto V = [&](uint32_t*& q, V3 v) { *q++ = F(v.x); *q++ = F(v.y); *q++ = F(v.z); };
    uint32_t* q = o;
    *q++ = (uint32_t)((uint64_t)A.tick_count & 0xffffffffu); *q++ = (uint32_t)((uint64_t)A.tick_count >> 32);
    *q++ = (uint32_t)G.score_line[0]; *q++ = (uint32_t)G.score_line[1]; *q++ = (uint32_t)G.last_touch_car_id;
    uint64_t pm = 0;
    for (int p = 0; p < 34; p++) if (A.pads[p].is_active) pm |= 1ull << p;
    *q++ = (uint32_t)pm; *q++ = (uint32_t)(pm >> 32);
    V(q, A.ball.b.pos * BT2UU); V(q, A.ball.b.vel * BT2UU); V(q, A.ball.b.angvel);
    *q++ = done ? 1u : 0u; *q++ = (uint32_t)NC;
    for (int k = 0; k < NC; k++) {
        const Car& c = A.cars[k];
        V(q, c.b.pos * BT2UU); V(q, col0(c.b.rot)); V(q, col1(c.b.rot)); V(q, col2(c.b.rot)); V(q, c.b.vel * BT2UU); V(q, c.b.angvel);
        *q++ = c.flags; *q++ = F(c.boost); *q++ = F(c.air_time_since_jump); *q++ = F(c.jump_time); *q++ = F(c.flip_time); *q++ = F(c.demo_respawn_timer);
        const bool tick = (c.flags & CF_BALLHIT_VALID) && c.bh_tick_hit == A.tick_count - 1;
        *q++ = ((touched_step && touched_step[k]) ? 1u : 0u) | (tick ? 2u : 0u);
        for (int i = 0; i < 8; i++) *q++ = (uint32_t)G.counters[k][i];
    }
}

// ---- GameEventTracker::Update ----------------------------------------------------------------------------
template <int NC>
RLG_HD bool ball_probably_going_in(const Arena<NC>& A, float max_time, int& goal_team) {  // Arena.cpp:827-863
    V3 bp = A.ball.b.pos * BT2UU, bv = A.ball.b.vel * BT2UU;
    if (fabsf(bv.y) < SIMD_EPS) return false;
    float dir = sgnf(bv.y);
    float goal_y = A.mut.goal_threshold_y * dir;   // MutatorConfig::goalBaseThresholdY (Arena.cpp:836)
    float dist = fabsf(bp.y - goal_y);
    float t = dist / fabsf(bv.y);
    if (t > max_time) return false;
    V3 grav = v3(A.mut.gravity_x, A.mut.gravity_y, A.mut.gravity_z);   // MutatorConfig::gravity (Arena.cpp:844)
    V3 ex = bp + (bv * t) + vdiv_rs(grav * t * t, 2.f);
    const float HW = 892.755f, GH = 642.775f;
    float margin = K::BALL_RADIUS * 0.1f + 0.f;
    if (ex.z > GH + margin) return false;
    if (fabsf(ex.x) > HW + margin) return false;
    goal_team = dir < 0 ? 0 : 1;  // RS_TEAM_FROM_Y(scoreDirSgn)
    return true;
}
template <int NC>
RLG_HD bool shooter_passer(const Arena<NC>& A, int team, int& shooter, bool find_passer, int& passer, int64_t max_shooter_ticks, int64_t max_passer_ticks) {
    shooter = -1; passer = -1;
    for (int k = 0; k < NC; k++) {
        const Car& c = A.cars[k];
        if ((k % 2) != team || !(c.flags & CF_BALLHIT_VALID)) continue;
        if (c.bh_tick_hit + max_shooter_ticks >= A.tick_count)
            if (shooter < 0 || c.bh_tick_hit > A.cars[shooter].bh_tick_hit) shooter = k;
    }
    if (shooter >= 0 && find_passer) {
        int64_t shoot_tick = A.cars[shooter].bh_tick_hit;
        for (int k = 0; k < NC; k++) {
            const Car& c = A.cars[k];
            if ((k % 2) != team || !(c.flags & CF_BALLHIT_VALID) || k == shooter) continue;
            if (c.bh_tick_hit + max_passer_ticks >= shoot_tick)
                if (passer < 0 || c.bh_tick_hit > A.cars[passer].bh_tick_hit) passer = k;
        }
    }
    return shooter >= 0;
}
template <int NC>
RLG_HD_T6A void event_tracker_update(const Arena<NC>& A, GymEnv<NC>& G) {
    const float tickrate = 1.f / TICK_DT;
    bool scored = fabsf(A.ball.b.pos.y * BT2UU) > (A.mut.goal_threshold_y + K::BALL_RADIUS);  // Arena.cpp:949-957 (goalBaseThresholdY + ballRadius)
    int64_t buc = A.ball_update_counter;
    bool ball_shot = G.tracker_flags & 1u, scored_last = G.tracker_flags & 4u; int shot_goal_team = (G.tracker_flags & 2u) ? 1 : 0;
    if (buc > G.last_ball_update_count) {
        int64_t delta_ticks = buc - G.last_ball_update_count;
        float delta_time = (float)delta_ticks * TICK_DT;
        if (RLG_UNLIKELY(scored && !scored_last)) {
            int sh, pa;
            int team = (-A.ball.b.pos.y) < 0 ? 0 : 1;
            if (shooter_passer(A, team, sh, true, pa, (int64_t)(4.0f * tickrate), (int64_t)(2.0f * tickrate))) {
                G.counters[sh][0]++;                 // matchGoals
                if (pa >= 0) G.counters[pa][2]++;    // matchAssists
            }
        } else if (!ball_shot) {
            if (G.shot_cooldown > 0) G.shot_cooldown = fmaxf(G.shot_cooldown - delta_time, 0.f);
            else {
                float sp2 = len2(A.ball.b.vel * BT2UU);
                if (RLG_UNLIKELY(sp2 >= 1750.f * 1750.f)) {
                    int goal_team;
                    if (ball_probably_going_in(A, 2.0f, goal_team)) {
                        int shooter_team = 1 - goal_team;
                        int64_t min_delay = (int64_t)(0.3f * tickrate);
                        int sh, pa;
                        if (shooter_passer(A, shooter_team, sh, true, pa, delta_ticks + min_delay, (int64_t)(2.0f * tickrate))) {
                            int64_t since = A.tick_count - A.cars[sh].bh_tick_hit;
                            if (since >= min_delay) {
                                ball_shot = true; shot_goal_team = goal_team; G.shot_cooldown = 1.0f;
                                G.counters[sh][3]++;               // matchShots
                                if (pa >= 0) G.counters[pa][4]++;  // matchShotPasses
                            }
                        }
                    }
                }
            }
        } else {
            int gt;
            if (!ball_probably_going_in(A, 2.0f, gt)) {
                int sv, un;
                if (shooter_passer(A, shot_goal_team, sv, false, un, delta_ticks, 0)) G.counters[sv][1]++;  // matchSaves
                ball_shot = false;
            }
        }
    } else if (buc == G.last_ball_update_count) {
        return;
    } else {
        ball_shot = false; scored = false; G.shot_cooldown = 0.f;  // ResetPersistentInfo, then _ballScoredLast = scored below
        scored = fabsf(A.ball.b.pos.y * BT2UU) > (A.mut.goal_threshold_y + K::BALL_RADIUS);
    }
    G.tracker_flags = (G.tracker_flags & ~7u) | (ball_shot ? 1u : 0u) | (shot_goal_team ? 2u : 0u) | (scored ? 4u : 0u);
    G.last_ball_update_count = buc;
}

// ---- DefaultOBS (OBSBuilders/DefaultOBS.cpp:3-55) -----------------------------------------------------------
RLG_HD V3 inv3(V3 v, bool inv) { return inv ? v3(-v.x, -v.y, v.z) : v; }
template <int NC>
RLG_HD int obs_size() { return 51 + 19 * NC; }
// row width for a config: DefaultOBS 51 + 19 * players, DefaultOBSPadded(m) 51 + 19 * 2m (self + m-1 mates + m opponents)
template <int NC>
RLG_HD int obs_size(const GymConfig& cfg) { return cfg.obs_max_players > 0 ? 51 + 38 * cfg.obs_max_players : 51 + 19 * players_per_env<NC>(cfg); }
constexpr int OBS_MAX_PADDED_PLAYERS = 4;

template <int NC>
RLG_HD float* obs_add_player(float* o, const Snapshot<NC>& S, int k, bool inv, const GymConfig& cfg) {
    V3 p = inv3(S.car_pos[k], inv), f = inv3(S.car_fwd[k], inv), u = inv3(S.car_up[k], inv), v = inv3(S.car_vel[k], inv), w = inv3(S.car_angvel[k], inv);
    *o++ = p.x * cfg.pos_coef[0]; *o++ = p.y * cfg.pos_coef[1]; *o++ = p.z * cfg.pos_coef[2];
    *o++ = f.x; *o++ = f.y; *o++ = f.z; *o++ = u.x; *o++ = u.y; *o++ = u.z;
    *o++ = v.x * cfg.vel_coef; *o++ = v.y * cfg.vel_coef; *o++ = v.z * cfg.vel_coef;
    *o++ = w.x * cfg.ang_vel_coef; *o++ = w.y * cfg.ang_vel_coef; *o++ = w.z * cfg.ang_vel_coef;
    *o++ = S.boost_frac[k]; *o++ = S.on_ground[k] ? 1.f : 0.f; *o++ = S.has_flip[k] ? 1.f : 0.f; *o++ = S.demoed[k] ? 1.f : 0.f;
    return o;
}
// DefaultOBS::BuildOBS (DefaultOBS.cpp:3-55).  With cfg.obs_max_players = m (DefaultOBSPadded.cpp:3-66) the teammate list is padded
// with zero blocks to m-1 entries, the opponent list to m, and each list is shuffled per observation; the reference shuffles with
// the process-wide std engine, here the permutation comes from the env's Philox stream keyed by (env, step, reset count, player).
template <int NC>
RLG_HD_T6B void build_obs(const Snapshot<NC>& S, int k, const float* prev_action8, const GymConfig& cfg, float* o, uint32_t env_id, uint32_t step, uint32_t resets) {
    bool inv = (k % 2) == 1;
    V3 bp = inv3(S.ball_pos, inv), bv = inv3(S.ball_vel, inv), bw = inv3(S.ball_angvel, inv);
    *o++ = bp.x * cfg.pos_coef[0]; *o++ = bp.y * cfg.pos_coef[1]; *o++ = bp.z * cfg.pos_coef[2];
    *o++ = bv.x * cfg.vel_coef; *o++ = bv.y * cfg.vel_coef; *o++ = bv.z * cfg.vel_coef;
    *o++ = bw.x * cfg.ang_vel_coef; *o++ = bw.y * cfg.ang_vel_coef; *o++ = bw.z * cfg.ang_vel_coef;
    for (int i = 0; i < 8; i++) *o++ = prev_action8[i];
    for (int i = 0; i < 34; i++) { int src = inv ? (33 - i) : i; *o++ = ((S.pads_active >> src) & 1ull) ? 1.f : 0.f; }
    o = obs_add_player(o, S, k, inv, cfg);
    constexpr int LIST = (NC > OBS_MAX_PADDED_PLAYERS ? NC : OBS_MAX_PADDED_PLAYERS) + 1;
    int mates[LIST], opps[LIST], nm = 0, no = 0;
    for (int j = 0; j < NC; j++) if (j != k && player_present(cfg, j)) { if ((j % 2) == (k % 2)) mates[nm++] = j; else opps[no++] = j; }   // state.players order
    if (cfg.obs_max_players > 0) {
        while (nm < cfg.obs_max_players - 1) mates[nm++] = -1;   // zero blocks (DefaultOBSPadded.cpp:46-54)
        while (no < cfg.obs_max_players) opps[no++] = -1;
        uint32_t r[4];
        philox4(cfg.seed_lo ^ 0x0B5E55EDu, cfg.seed_hi, env_id, step, (resets << 8) | (uint32_t)k, r);
        // Fisher-Yates from the back, one 16-bit draw per swap: two draws per Philox word, words 0 and 2 for the mates, 1 and 3 for the opponents
        uint32_t bits = r[0]; int used = 0;
        for (int i = nm - 1; i > 0; i--) {
            if (used == 2) bits = r[2];
            int j = (int)((bits & 0xffffu) % (uint32_t)(i + 1)); bits >>= 16; used++;
            int t = mates[i]; mates[i] = mates[j]; mates[j] = t;
        }
        bits = r[1]; used = 0;
        for (int i = no - 1; i > 0; i--) {
            if (used == 2) bits = r[3];
            int j = (int)((bits & 0xffffu) % (uint32_t)(i + 1)); bits >>= 16; used++;
            int t = opps[i]; opps[i] = opps[j]; opps[j] = t;
        }
    }
    for (int i = 0; i < nm; i++) { if (mates[i] >= 0) o = obs_add_player(o, S, mates[i], inv, cfg); else for (int q = 0; q < 19; q++) *o++ = 0.f; }
    for (int i = 0; i < no; i++) { if (opps[i] >= 0) o = obs_add_player(o, S, opps[i], inv, cfg); else for (int q = 0; q < 19; q++) *o++ = 0.f; }
}

// ---- rewards ---------------------------------------------------------------------------------------------
RLG_HD V3 rs_normalized(V3 v) {  // RocketSim Vec::Normalized (MathTypes.h:88-95)
    float l2 = len2(v);
    float l = l2 > 0 ? sqrtf(l2) : 0.f;
    if (l > SIMD_EPS * SIMD_EPS) return vdiv_rs(v, l);
    return v3(0, 0, 0);
}
template <int NC>
RLG_HD void event_values(const Snapshot<NC>& S, const GymEnv<NC>& G, int k, float* v) {  // EventReward::ExtractValues
    int team = k % 2;
    v[0] = (float)G.counters[k][0]; v[1] = (float)G.score_line[team]; v[2] = (float)G.score_line[1 - team]; v[3] = (float)G.counters[k][2];
    v[4] = S.touched[k] ? 1.f : 0.f; v[5] = (float)G.counters[k][3]; v[6] = (float)G.counters[k][4]; v[7] = (float)G.counters[k][1];
    v[8] = (float)G.counters[k][6]; v[9] = S.demoed[k] ? 1.f : 0.f; v[10] = S.boost_frac[k];
}
// powf as the reference's host libm returns it: glibc's algorithm restated (rl_libm.h rl_powf; its result is within 0.82 ulp, not always the correctly rounded
// one).  The device library's powf is a float algorithm good to 1 ulp, which showed as last-bit differences in SaveBoostReward / TouchBallReward on the random
// 2v2 / 3v3 rollouts (round 2); the double pow rounded to float that replaced it is correctly rounded, which glibc's is not quite (round 6: one reward of 120
// live rollouts an ulp off).  A few hundred cycles per player and STEP -- nothing next to the 8 ticks before it; kept out of line.
RLG_HD_COLD float libm_powf(float x, float y) { return rl_powf(x, y); }
template <int NC>
RLG_HD_T6B void compute_rewards(const Snapshot<NC>& S, GymEnv<NC>& G, const GymConfig& cfg, float* rew) {
    for (int k = 0; k < NC; k++) rew[k] = 0.f;
    for (int t = 0; t < cfg.n_terms; t++) {
        const RewardTerm& T = cfg.terms[t];
        for (int k = 0; k < NC; k++) {
            if (!player_present(cfg, k)) continue;
            float r = 0.f;
            switch (T.kind) {
                case RW_EVENT: {
                    float nv[RLGPU_NUM_EVENT_VALS]; event_values(S, G, k, nv);
                    for (int i = 0; i < RLGPU_NUM_EVENT_VALS; i++) { r += fmaxf(nv[i] - G.event_last[k][i], 0.f) * cfg.event_weights[i]; G.event_last[k][i] = nv[i]; }
                } break;
                case RW_VELOCITY: r = len(S.car_vel[k]) / 2300.f * (1 - 2 * (T.p0 != 0.f ? 1 : 0)); break;
                case RW_SAVE_BOOST: r = clampf(libm_powf(S.boost_frac[k], T.p0), 0.f, 1.f); break;
                case RW_VEL_BALL_TO_GOAL: {
                    bool orange_goal = (k % 2) == 0;
                    if (T.p0 != 0.f) orange_goal = !orange_goal;
                    V3 target = v3(0.f, orange_goal ? 6000.f : -6000.f, 642.775f / 2);
                    V3 dir = rs_normalized(target - S.ball_pos);
                    r = dot(dir, vdiv_rs(S.ball_vel, 6000.f));
                } break;
                case RW_VEL_PLAYER_TO_BALL: {
                    V3 dir = rs_normalized(S.ball_pos - S.car_pos[k]);
                    r = dot(dir, vdiv_rs(S.car_vel[k], 2300.f));
                } break;
                case RW_FACE_BALL: {
                    V3 dir = rs_normalized(S.ball_pos - S.car_pos[k]);
                    r = dot(S.car_fwd[k], dir);
                } break;
                case RW_TOUCH_BALL: r = S.touched[k] ? libm_powf((S.ball_pos.z + 92.75f) / (92.75f * 2), T.p0) : 0.f; break;
                default: break;
            }
            rew[k] += r * T.weight;
        }
    }
    if (cfg.zero_sum) {
        // the team sums run over GameState::players (ZeroSumReward.cpp:9-13), i.e. in the arena's car order: with three players per team the
        // order of a float sum shows in the last bit
        float avg[2] = {0.f, 0.f}; int cnt[2] = {0, 0};
        for (int r = 0; r < NC; r++) { const int k = S.car_order ? (int)((S.car_order >> (4 * r)) & 15u) - 1 : r; if (player_present(cfg, k)) { cnt[k % 2]++; avg[k % 2] += rew[k]; } }
        for (int t = 0; t < 2; t++) avg[t] /= (float)(cnt[t] > 1 ? cnt[t] : 1);
        for (int k = 0; k < NC; k++) { if (!player_present(cfg, k)) continue; int t = k % 2; rew[k] = rew[k] * (1 - cfg.team_spirit) + (avg[t] * cfg.team_spirit) - (avg[1 - t] * cfg.opp_scale); }
    }
}

// ---- terminal conditions (Match::IsDone short-circuit, Match.cpp:32-38) ------------------------------------------
template <int NC>
RLG_HD bool compute_done(const Snapshot<NC>& S, GymEnv<NC>& G, const GymConfig& cfg) {
    for (int i = 0; i < cfg.n_conds; i++) {
        if (cfg.conds[i] == TC_NO_TOUCH) {
            bool any = false;
            for (int k = 0; k < NC; k++) any = any || S.touched[k];
            if (any) { G.no_touch_steps = 0; }
            else { G.no_touch_steps++; if (G.no_touch_steps >= cfg.no_touch_max_steps) return true; }
        } else if (cfg.conds[i] == TC_GOAL_SCORE) {
            if (fabsf(S.ball_pos.y) > K::GOAL_THRESHOLD_Y + K::BALL_RADIUS) return true;
        }
    }
    return false;
}

// ---- state setters ---------------------------------------------------------------------------------------------
struct Rng {
    uint32_t s0, s1, stream, ctr, sub; uint32_t buf[4]; int have;
    RLG_HD uint32_t next() {
        if (have == 0) { philox4(s0, s1, stream, ctr, sub, buf); sub++; have = 4; }
        return buf[4 - (have--)];
    }
    RLG_HD float uni(float lo, float hi) { return lo + u32_to_unit(next()) * (hi - lo); }
    // three draws in x,y,z order (function-argument evaluation order is unspecified in C++: never draw inside an argument list)
    RLG_HD V3 uni3(float lx, float hx, float ly, float hy, float lz, float hz) { float x = uni(lx, hx); float y = uni(ly, hy); float z = uni(lz, hz); return v3(x, y, z); }
};
// (euler_to_rot -- Angle::ToRotMat -- lives in arena_car.h, next to Car::Respawn, its other user)
RLG_HD void car_set_fresh(Car& c) {  // Car::SetState(CarState()) semantics: carried wheel values / controls survive
    Car n = c;
    n.flags = CF_ON_GROUND; n.flip_rel_torque = v3(0, 0, 0);
    n.jump_time = n.flip_time = n.air_time = n.air_time_since_jump = 0.f;
    n.boost = K::BOOST_SPAWN_AMOUNT; n.time_spent_boosting = n.supersonic_time = n.handbrake_val = 0.f;
    n.auto_flip_timer = n.auto_flip_torque_scale = 0.f; n.world_contact_normal = v3(0, 0, 0);
    n.car_contact_other = 0; n.car_contact_cooldown = 0.f; n.demo_respawn_timer = 0.f;
    n.bh_rel_pos = n.bh_ball_pos = n.bh_extra_hit_vel = v3(0, 0, 0); n.bh_tick_hit = -1; n.bh_tick_extra = -1;
    n.last.throttle = n.last.steer = n.last.pitch = n.last.yaw = n.last.roll = 0.f; n.last.jump = n.last.boost = n.last.handbrake = false;
    n.vel_impulse_cache = v3(0, 0, 0);
    n.b.vel = v3(0, 0, 0); n.b.angvel = v3(0, 0, 0);
    c = n;
}
// BoostPad::SetState(BoostPadState()) for every pad
template <int NC>
RLG_HD void reset_pads(Arena<NC>& A) {
    for (int p = 0; p < 34; p++) { A.pads[p].cooldown = 0.f; A.pads[p].is_active = true; A.pads[p].prev_locked = 0; A.pads[p].cur_locked = 0; }
}
// what a built-in setter reads of the gym's configuration, BY VALUE: reset_state is a real call, and a reference into a kernel's argument struct that
// escapes into one makes the compiler keep the whole struct in scratch memory (every cfg.x of the step loop then is a scratch load instead of a scalar load)
struct SetterCfg { int32_t setter_kind, rand_ball_speed, rand_car_speed, cars_on_ground; uint32_t seed_lo, seed_hi; };
RLG_HD SetterCfg setter_cfg(const GymConfig& c) { SetterCfg s; s.setter_kind = c.setter_kind; s.rand_ball_speed = c.rand_ball_speed; s.rand_car_speed = c.rand_car_speed; s.cars_on_ground = c.cars_on_ground; s.seed_lo = c.seed_lo; s.seed_hi = c.seed_hi; return s; }
// The setters' draws come from one of two sources.  `Rng` above: the env's Philox stream (product).  `RefRng`: the reference thread's engine
// (Arena::ref_engine != 0, parity tests), with RocketSim's RandFloat and -- where the reference draws inside an argument list -- in the order in which
// its compiled code evaluates them: g++ evaluates the arguments of `Vec(RandFloat(x), RandFloat(y), RandFloat(z))` (Math.cpp:7-13) and of
// `Angle(RandFloat(yaw), RandFloat(pitch), RandFloat(roll))` (RandomState.cpp:44) RIGHT TO LEFT, and `RandNormVec() * RandFloat(0, s)` left operand
// first (pinned against the reference as compiled by oracle/Makefile: tests/golden/setter_golden.npz).
struct RefRng {
    RefEngine e;
    static constexpr bool REF = true;
    RLG_HD float uni(float lo, float hi) { return e.uni(lo, hi); }
    RLG_HD V3 uni3(float lx, float hx, float ly, float hy, float lz, float hz) { float z = uni(lz, hz); float y = uni(ly, hy); float x = uni(lx, hx); return v3(x, y, z); }
    RLG_HD void ypr(float& yaw, float& pitch, float& roll) { roll = uni(-PI_F, PI_F); pitch = uni(-PI_F / 2, PI_F / 2); yaw = uni(-PI_F, PI_F); }
    RLG_HD V3 dir_times_speed(float max_speed) { const float sp = uni(0, max_speed); const V3 d = rs_normalized(uni3(-1, 1, -1, 1, -1, 1)); return d * sp; }   // `RandNormVec() * RandFloat(0, s)`: the right operand first
    RLG_HD void kickoff_order(int (&order)[5]) { e.shuffle5(order); }
};
struct PhiloxRng : Rng {
    static constexpr bool REF = false;
    RLG_HD void ypr(float& yaw, float& pitch, float& roll) { yaw = uni(-PI_F, PI_F); pitch = uni(-PI_F / 2, PI_F / 2); roll = uni(-PI_F, PI_F); }
    RLG_HD V3 dir_times_speed(float max_speed) { const V3 d = rs_normalized(uni3(-1, 1, -1, 1, -1, 1)); return d * uni(0, max_speed); }
    RLG_HD void kickoff_order(int (&order)[5]) { for (int i = 4; i > 0; i--) { int j = (int)(next() % (uint32_t)(i + 1)); int t = order[i]; order[i] = order[j]; order[j] = t; } }
};
template <int NC, class RNG>
RLG_HD void reset_state_with(Arena<NC>& A, const SetterCfg cfg, RNG& rng) {
    A.ball.vel_impulse_cache = v3(0, 0, 0); A.ball_update_counter = 0;
    // The boost pads: both built-in setters reset them BEFORE they build the new episode's first GameState -- KickoffState through
    // Arena::ResetToRandomKickoff (Arena.cpp:209-210), RandomState because its first statement is that same call (RandomState.cpp:11) -- so the
    // first observation of the new episode shows all 34 pads active (tests/golden/padreset_golden.npz).  Only a USER setter that leaves the pads
    // alone shows the previous episode's pads there: Match::ResetState resets them after the setter returned (Match.cpp:55-69; the reset_pads of
    // gym_episode_reset, the only one that runs on the host-setter path).
    reset_pads(A);
    // Arena::ResetToRandomKickoff (Arena.cpp:112-216).  RandomState begins with the same call (RandomState.cpp:11): every car and the ball are placed
    // anew right after, but the shuffle's draws are taken from the thread's engine first -- skipped on the env's own stream, which nobody compares draw for draw
    if (cfg.setter_kind == SS_KICKOFF || RNG::REF) {
        const float SX[5] = {-2048, 2048, -256, 256, 0}, SY[5] = {-2560, -2560, -3840, -3840, -4608};
        // yawAng = (float)(M_PI_4 * k) (RLConst.h:297-303: a double product rounded once); orange: `angle.yaw += M_PI` in double, rounded once (Arena.cpp:186)
        const double PI_4_D = 0.78539816339744830962, PI_D = 3.14159265358979323846;
        const float SYAW[5] = {(float)(PI_4_D * 1), (float)(PI_4_D * 3), (float)(PI_4_D * 2), (float)(PI_4_D * 2), (float)(PI_4_D * 2)};
        int order[5] = {0, 1, 2, 3, 4};
        rng.kickoff_order(order);
        // position i goes to the i-th car of each team in the order in which the arena's car set lists them (Arena.cpp:149-151: blueCars / orangeCars are
        // filled by iterating `_cars`)
        int per_team[2] = {0, 0};
        for (int r = 0; r < NC; r++) {
            const int k = car_at_rank(A, r); const bool blue = (k % 2) == 0;
            const int i = per_team[blue ? 0 : 1]++;
            int sl = order[i < 5 ? i : 4];
            Car& c = A.cars[k]; car_set_fresh(c);
            V3 pos = v3(SX[sl], SY[sl], K::CAR_SPAWN_REST_Z); float yaw = SYAW[sl];
            if (!blue) { pos = pos * v3(-1, -1, 1); yaw = (float)((double)yaw + PI_D); }
            c.b.pos = pos * UU2BT; c.b.rot = euler_to_rot(yaw, 0.f, 0.f);
        }
        A.ball.b.pos = v3(0, 0, K::BALL_REST_Z) * UU2BT; A.ball.b.vel = v3(0, 0, 0); A.ball.b.angvel = v3(0, 0, 0);
        A.ball.b.rot = m3_identity();      // a BallState's default rotMat
    }
    if (cfg.setter_kind != SS_KICKOFF) {
        // RandomState::ResetState (StateSetters/RandomState.cpp:8-61)
        const float X_MAX = 3500, Y_MAX = 4000, Z_MAX = 1820, CAR_Z_MIN = 150;
        V3 bp = rng.uni3(-X_MAX, X_MAX, -Y_MAX, Y_MAX, 92.75f, Z_MAX);
        V3 bv = v3(0, 0, 0), bw = v3(0, 0, 0);
        if (cfg.rand_ball_speed) {
            bv = rng.dir_times_speed(4000);
            bw = rng.uni3(-4, 4, -4, 4, -4, 4);
        }
        A.ball.b.pos = bp * UU2BT; A.ball.b.vel = bv * UU2BT; A.ball.b.angvel = bw;
        A.ball.b.rot = m3_identity();
        for (int r = 0; r < NC; r++) {
            const int k = RNG::REF ? car_at_rank(A, r) : r;     // `for (Car* car : arena->_cars)`: the arena's car order (only the reference's engine cares)
            Car& c = A.cars[k]; car_set_fresh(c);
            V3 pos = rng.uni3(-X_MAX, X_MAX, -Y_MAX, Y_MAX, CAR_Z_MIN, Z_MAX);
            V3 vel = v3(0, 0, 0), av = v3(0, 0, 0);
            if (cfg.rand_car_speed) {
                if (RNG::REF) (void)rng.uni3(-1, 1, -1, 1, -1, 1);   // `randVelDir` (RandomState.cpp:39): drawn and never used
                vel = rng.dir_times_speed(K::CAR_MAX_SPEED);
                V3 d2 = rs_normalized(rng.uni3(-1, 1, -1, 1, -1, 1));
                av = d2 * 5.5f;
            }
            float yaw, pitch, roll; rng.ypr(yaw, pitch, roll);
            bool on_ground = cfg.cars_on_ground ? true : (rng.uni(0, 1) > 0.5f);
            if (on_ground) { pos.z = 17; pitch = roll = 0; vel.z = 0; av = v3(0, 0, 0); }
            c.b.pos = pos * UU2BT; c.b.rot = euler_to_rot(yaw, pitch, roll); c.b.vel = vel * UU2BT; c.b.angvel = av;
            c.boost = rng.uni(0, 100);
        }
    }
    arena_finish_load(A);
}
template <int NC>
RLG_HD_NOINLINE void reset_state(Arena<NC>& A, GymEnv<NC>& G, const SetterCfg cfg, uint32_t env_id) {
    if (RLG_UNLIKELY(A.ref_engine != 0u)) {
        RefRng rng; rng.e.x = A.ref_engine;
        G.reset_count++;
        reset_state_with(A, cfg, rng);
        A.ref_engine = rng.e.x;
        return;
    }
    PhiloxRng rng; rng.s0 = cfg.seed_lo; rng.s1 = cfg.seed_hi; rng.stream = env_id; rng.ctr = G.reset_count; rng.sub = 0; rng.have = 0;
    G.reset_count++;
    reset_state_with(A, cfg, rng);
}
// the orange slots of a one-team env (also after a host state setter, which knows nothing about them)
template <int NC>
RLG_HD void park_absent_players(Arena<NC>& A, const GymConfig& cfg) {
    if (!cfg.one_team) return;
    for (int k = 1; k < NC; k += 2) {
        Car& c = A.cars[k];
        c.flags = CF_IS_DEMOED | CF_ABSENT; c.demo_respawn_timer = ABSENT_RESPAWN_TIMER;
        c.b.pos = v3(0, 0, -10000.f) * UU2BT; c.b.vel = v3(0, 0, 0); c.b.angvel = v3(0, 0, 0); c.vel_impulse_cache = v3(0, 0, 0);
        c.bh_tick_hit = -1; c.bh_tick_extra = -1; c.boost = 0.f;
    }
}

// Gym::Reset bookkeeping after the state setter ran (Gym.cpp:58-66, Match.cpp:4-10, GameState ctor)
template <int NC>
RLG_HD void gym_episode_reset(Arena<NC>& A, GymEnv<NC>& G, const GymConfig& cfg, Snapshot<NC>& S) {
    park_absent_players(A, cfg);
    G.score_line[0] = G.score_line[1] = 0; G.last_touch_car_id = -1; G.last_tick_count = 0;
    for (int k = 0; k < NC; k++) for (int q = 0; q < 8; q++) G.counters[k][q] = 0;
    take_snapshot(A, G, S);  // GameState(arena): lastTickCount was 0 -> tickSkip = tickCount
    // Match::ResetState resets the pads only now (Match.cpp:55-69: `newState = stateSetter->ResetState(arena)` first, `pad->SetState({})` after):
    // the new episode's first GameState -- and with it the first observation -- still shows the pads as the previous episode left them, unless
    // the setter itself reset them (KickoffState through Arena::ResetToRandomKickoff)
    reset_pads(A);
    G.no_touch_steps = 0;
    for (int k = 0; k < NC; k++) { G.prev_action_idx[k] = -1; event_values(S, G, k, G.event_last[k]); }
    G.tracker_flags &= ~(1u | 4u); G.shot_cooldown = 0.f;  // ResetPersistentInfo (GameEventTracker.cpp:160-165)
    G.episode_steps = 0;
}

// ---- Gym::Step for one env ---------------------------------------------------------------------------------------
// actions: NC indices into the action table (slot order).  Writes, in slot order:
//   reward[NC], done, and next_obs[NC][D] = the observation the policy sees next (post-reset when done).
// The step is cut at the tick boundaries (begin | tick 1 | after_first_tick | ticks 2..tickSkip | end) so that the device
// kernel can run the ticks with a whole wavefront (rlgpu_env.hip) while the host build calls arena_tick() in between.
// The observation of a step is built right after its first tick, from the snapshot taken there (Gym.cpp:81-93 builds it from that same
// GameState after the remaining ticks): the snapshot then does not have to outlive the other ticks -- on the device it sits in the
// env's TickWork area, which is dead between ticks, instead of per-lane scratch memory that stays allocated across the whole step
// (measured: -25 % HBM reads and -5 % HBM writes of the collection kernel; the rest of its scratch traffic are per-tick callee frames).
template <int NC>
RLG_HD void gym_step_begin(Arena<NC>& A, GymEnv<NC>& G, const GymConfig& cfg, const float* action_table, const int32_t* actions) {
    // Match::ParseActions: demoed players (per the PREVIOUS snapshot) get a zero action (Match.cpp:44-52)
    uint32_t snap_demoed = (G.tracker_flags >> 8) & 0xffu;
    for (int k = 0; k < NC; k++) {
        int idx = player_present(cfg, k) ? actions[agent_row(cfg, k)] : -1;
        bool zero = (snap_demoed >> k) & 1u;
        if (idx < 0 || idx >= cfg.n_actions) zero = true;
        float pa[8];
        for (int i = 0; i < 8; i++) pa[i] = zero ? 0.f : action_table[idx * 8 + i];
        G.prev_action_idx[k] = zero ? -1 : idx;
        Controls& c = A.cars[k].ctl;  // Action -> CarControls (Action.h:36-46)
        c.throttle = pa[0]; c.steer = pa[1]; c.pitch = pa[2]; c.yaw = pa[3]; c.roll = pa[4];
        c.jump = pa[5] == 1.f; c.boost = pa[6] == 1.f; c.handbrake = pa[7] == 1.f;
    }
}

// after arena->Step(tickSkip - actionDelay) = 1 tick: `ev` holds the bump callbacks of that tick.  S: scratch for the GameState of
// this step.  Writes reward[NC] and next_obs[NC][D] (the latter is replaced by gym_step_end when the episode ended); returns done.
template <int NC>
RLG_HD bool gym_step_after_first_tick(Arena<NC>& A, GymEnv<NC>& G, const GymConfig& cfg, const TickEvents& ev, const float* action_table, uint32_t env_id,
                                      float* reward, float* next_obs, size_t obs_row_stride, Snapshot<NC>& S) {
    // bump callbacks that fired during this first tick land in the snapshot (later ones are lost: Gym.cpp:84-96)
    for (int k = 0; k < NC; k++) {
        if (ev.bump_mask & (1u << k)) G.counters[k][5]++;
        if (ev.bump_mask & (1u << (8 + k))) G.counters[k][6]++;
    }
    event_tracker_update(A, G);
    take_snapshot(A, G, S);
    uint32_t dm = 0; for (int k = 0; k < NC; k++) if (S.demoed[k]) dm |= (1u << k);
    G.tracker_flags = (G.tracker_flags & ~0xff00u) | (dm << 8);
    const bool done = compute_done(S, G, cfg);
    float rew[NC];
    compute_rewards(S, G, cfg, rew);
    for (int k = 0; k < NC; k++) if (player_present(cfg, k)) reward[agent_row(cfg, k)] = rew[k];
    if (!done) {   // the step counter the obs builder keys its shuffle with is the one gym_step_end is about to write
        for (int k = 0; k < NC; k++) {
            if (!player_present(cfg, k)) continue;
            float pa[8];
            const int idx = G.prev_action_idx[k];
            for (int i = 0; i < 8; i++) pa[i] = idx < 0 ? 0.f : action_table[idx * 8 + i];
            build_obs(S, k, pa, cfg, next_obs + (size_t)agent_row(cfg, k) * obs_row_stride, env_id, G.episode_steps + 1, G.reset_count);
        }
    }
    return done;
}

template <int NC>
RLG_HD void gym_step_end(Arena<NC>& A, GymEnv<NC>& G, const GymConfig& cfg, uint32_t env_id, float* next_obs, size_t obs_row_stride, bool done, Snapshot<NC>& S) {
    G.episode_steps++;
    if (RLG_UNLIKELY(done && !cfg.host_resets)) {   // GameInst::Step: the recorded next observation is the first one of the new episode (GameInst.cpp:27-32)
        reset_state(A, G, setter_cfg(cfg), env_id);
        gym_episode_reset(A, G, cfg, S);
        G.tracker_flags &= ~0xff00u;
        const float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < NC; k++) if (player_present(cfg, k)) build_obs(S, k, zero, cfg, next_obs + (size_t)agent_row(cfg, k) * obs_row_stride, env_id, G.episode_steps, G.reset_count);
    }
}

template <int NC, int BIG>
RLG_HD void gym_step_env(Arena<NC>& A, GymEnv<NC>& G, const GymConfig& cfg, MeshView mesh, const float* action_table,
                         const int32_t* actions, uint32_t env_id, float* next_obs, size_t obs_row_stride, float* reward, int32_t* done_out, TickWork<NC, BIG>& W) {
    Snapshot<NC> S;
    gym_step_begin(A, G, cfg, action_table, actions);
    const uint32_t seed = tick_seed(cfg);
    TickEvents ev; ev.bump_mask = 0;
    arena_tick(A, mesh, seed, env_id, ev, W);
    const bool done = gym_step_after_first_tick(A, G, cfg, ev, action_table, env_id, reward, next_obs, obs_row_stride, S);
    *done_out = done ? 1 : 0;
    TickEvents ev2;
    for (int t = 1; t < cfg.tick_skip; t++) { ev2.bump_mask = 0; arena_tick(A, mesh, seed, env_id, ev2, W); }
    gym_step_end(A, G, cfg, env_id, next_obs, obs_row_stride, done, S);
}

// Gym::Reset for one env: state setter + bookkeeping + first observation
template <int NC>
RLG_HD void gym_reset_env(Arena<NC>& A, GymEnv<NC>& G, const GymConfig& cfg, uint32_t env_id, float* obs, size_t obs_row_stride, bool run_setter) {
    if (run_setter) reset_state(A, G, setter_cfg(cfg), env_id);
    Snapshot<NC> S;
    gym_episode_reset(A, G, cfg, S);
    G.tracker_flags &= ~0xff00u;
    float zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (obs) for (int k = 0; k < NC; k++) if (player_present(cfg, k)) build_obs(S, k, zero, cfg, obs + (size_t)agent_row(cfg, k) * obs_row_stride, env_id, G.episode_steps, G.reset_count);
}

}  // namespace rlg
