// arena_io.h — moving one env between its three representations:
//   (1) RlgpuArenaState   host AoS exchange struct in uu            (include/rlgpu_state.h)
//   (2) Arena<NC>+GymEnv  the working copy a lane steps             (arena_types.h / arena_gym.h)
//   (3) SoA words         the resident device layout [word][env]    (32-bit words, coalesced across lanes)
// (3) is defined by ONE field visitor, `arena_visit`, so load and store can never disagree.
#pragma once
#include "arena_body.h"
#include "../../include/rlgpu_state.h"

namespace rlg {

// gym-level carried state of one env (see RlgpuGymState for the reference fields it restates)
template <int NC>
struct GymEnv {
    int32_t score_line[2];
    int32_t last_touch_car_id;
    int64_t last_tick_count;
    int32_t no_touch_steps;
    float shot_cooldown;
    uint32_t tracker_flags;  // bit0 ball_shot, bit1 ball_shot_goal_team, bit2 ball_scored_last
    int64_t last_ball_update_count;
    int32_t counters[NC][8];  // goals saves assists shots shot_passes bumps demos boost_pickups
    float event_last[NC][RLGPU_NUM_EVENT_VALS];
    int32_t prev_action_idx[NC];   // index into the action table; -1 = zero action (reset / demoed)
    uint32_t episode_steps;
    uint32_t reset_count;          // RNG counter for state setters
};

RLG_HD float u2f(uint32_t u) { union { float f; uint32_t u; } x; x.u = u; return x.f; }

// word accessors used by the visitor
struct WordReader {
    const uint32_t* base; size_t stride; size_t idx;
    RLG_HD uint32_t next() { uint32_t v = base[idx * stride]; idx++; return v; }
    RLG_HD void f(float& x) { x = u2f(next()); }
    RLG_HD void i(int32_t& x) { x = (int32_t)next(); }
    RLG_HD void u(uint32_t& x) { x = next(); }
    RLG_HD void l(int64_t& x) { uint32_t lo = next(), hi = next(); x = (int64_t)(((uint64_t)hi << 32) | lo); }
    RLG_HD void v(V3& x) { f(x.x); f(x.y); f(x.z); }
};
struct WordWriter {
    uint32_t* base; size_t stride; size_t idx;
    RLG_HD void put(uint32_t v) { base[idx * stride] = v; idx++; }
    RLG_HD void f(float& x) { put(f2u(x)); }
    RLG_HD void i(int32_t& x) { put((uint32_t)x); }
    RLG_HD void u(uint32_t& x) { put(x); }
    RLG_HD void l(int64_t& x) { put((uint32_t)((uint64_t)x & 0xffffffffu)); put((uint32_t)((uint64_t)x >> 32)); }
    RLG_HD void v(V3& x) { f(x.x); f(x.y); f(x.z); }
};
struct WordCounter {
    size_t idx = 0;
    RLG_HD void f(float&) { idx++; }
    RLG_HD void i(int32_t&) { idx++; }
    RLG_HD void u(uint32_t&) { idx++; }
    RLG_HD void l(int64_t&) { idx += 2; }
    RLG_HD void v(V3&) { idx += 3; }
};

RLG_HD uint32_t pack_ctl(const Controls& c) { return (c.jump ? 1u : 0u) | (c.boost ? 2u : 0u) | (c.handbrake ? 4u : 0u); }
RLG_HD void unpack_ctl(Controls& c, uint32_t b) { c.jump = b & 1u; c.boost = b & 2u; c.handbrake = b & 4u; }

template <int NC, class IO>
RLG_HD void arena_visit(Arena<NC>& A, GymEnv<NC>& G, IO& io) {
    io.l(A.tick_count); io.l(A.ball_update_counter);
    io.v(A.ball.b.pos); io.v(A.ball.b.vel); io.v(A.ball.b.angvel); io.v(A.ball.vel_impulse_cache);
    io.v(A.ball.b.rot.r0); io.v(A.ball.b.rot.r1); io.v(A.ball.b.rot.r2);   // BallState::rotMat: constant under ArenaConfig::noBallRot, but whatever a state setter made it
    for (int k = 0; k < NC; k++) {
        Car& c = A.cars[k];
        io.v(c.b.pos); io.v(c.b.rot.r0); io.v(c.b.rot.r1); io.v(c.b.rot.r2); io.v(c.b.vel); io.v(c.b.angvel);
        io.u(c.flags); io.v(c.flip_rel_torque);
        io.f(c.jump_time); io.f(c.flip_time); io.f(c.air_time); io.f(c.air_time_since_jump); io.f(c.boost);
        io.f(c.time_spent_boosting); io.f(c.supersonic_time); io.f(c.handbrake_val); io.f(c.auto_flip_timer); io.f(c.auto_flip_torque_scale);
        io.v(c.world_contact_normal); io.i(c.car_contact_other); io.f(c.car_contact_cooldown); io.f(c.demo_respawn_timer);
        io.v(c.bh_rel_pos); io.v(c.bh_ball_pos); io.v(c.bh_extra_hit_vel); io.l(c.bh_tick_hit); io.l(c.bh_tick_extra);
        io.f(c.last.throttle); io.f(c.last.steer); io.f(c.last.pitch); io.f(c.last.yaw); io.f(c.last.roll);
        io.f(c.ctl.throttle); io.f(c.ctl.steer); io.f(c.ctl.pitch); io.f(c.ctl.yaw); io.f(c.ctl.roll);
        uint32_t bits = pack_ctl(c.last) | (pack_ctl(c.ctl) << 3);
        io.u(bits);
        Controls tmp; unpack_ctl(tmp, bits & 7u); c.last.jump = tmp.jump; c.last.boost = tmp.boost; c.last.handbrake = tmp.handbrake;
        unpack_ctl(tmp, (bits >> 3) & 7u); c.ctl.jump = tmp.jump; c.ctl.boost = tmp.boost; c.ctl.handbrake = tmp.handbrake;
        io.v(c.vel_impulse_cache);
        for (int w = 0; w < 4; w++) io.f(c.extra_pushback[w]);
        io.f(c.steer_angle); io.f(c.engine_force); io.f(c.brake);
        for (int w = 0; w < 4; w++) io.f(c.lat_friction[w]);
        for (int w = 0; w < 4; w++) io.f(c.long_friction[w]);
        // hidden state, not part of the exchange struct: the basis of a DEMOLISHED car's rigid body (car_ghost_rot; any other car: its
        // world inverse inertia, recomputed on load)
        io.v(c.b.inv_inertia_w.r0); io.v(c.b.inv_inertia_w.r1); io.v(c.b.inv_inertia_w.r2);
    }
    for (int p = 0; p < 34; p++) {
        Pad& pd = A.pads[p];
        io.f(pd.cooldown);
        uint32_t bits = (pd.is_active ? 1u : 0u) | ((uint32_t)pd.prev_locked << 1);
        io.u(bits);
        pd.is_active = bits & 1u; pd.prev_locked = (int8_t)(bits >> 1);
    }
    io.i(G.score_line[0]); io.i(G.score_line[1]); io.i(G.last_touch_car_id); io.l(G.last_tick_count); io.i(G.no_touch_steps);
    io.f(G.shot_cooldown);
    {   // two words: the event tracker's three flags with the per-car loop order above them, and which players the step's GameState showed as demolished
        // (Match::ParseActions gives those a zero action in the NEXT step, Match.cpp:44-52).  Until round 6 the second was not resident: it lived in the
        // fused collection kernel's LDS copy for the length of a launch and was lost between launches and between k_env_step calls, so a demolished
        // player's action was applied (and shown as its previous action) where the reference zeroes it -- found by the mutator fixture's random 2v2 rollout
        // with ON_CONTACT demolitions; the older demolition fixture's victim happened to idle.
        uint32_t sd = (G.tracker_flags >> 8) & 0xffu;
        uint32_t tf = (G.tracker_flags & 0xffu) | (A.car_order << 8);
        io.u(tf); io.u(sd);
        G.tracker_flags = (tf & 0xffu) | ((sd & 0xffu) << 8); A.car_order = tf >> 8;
    }
    io.l(G.last_ball_update_count);
    for (int b = 0; b <= NC; b += 2) {     // the broadphase history, two bodies per word
        uint32_t w = (uint32_t)A.bp_hist[b] | (b + 1 <= NC ? (uint32_t)A.bp_hist[b + 1] << 16 : 0u);
        io.u(w);
        A.bp_hist[b] = (uint16_t)(w & 0xffffu); if (b + 1 <= NC) A.bp_hist[b + 1] = (uint16_t)(w >> 16);
    }
    for (int k = 0; k < NC; k++) {
        for (int q = 0; q < 8; q++) io.i(G.counters[k][q]);
        for (int q = 0; q < RLGPU_NUM_EVENT_VALS; q++) io.f(G.event_last[k][q]);
        io.i(G.prev_action_idx[k]);
    }
    io.u(G.episode_steps); io.u(G.reset_count);
    io.u(A.ref_engine);
    {   // MutatorConfig's run-time scalars (MUTATOR_WORDS)
        Mutators& m = A.mut;
        io.f(m.gravity_z); io.f(m.boost_accel_ground); io.f(m.boost_accel_air); io.f(m.boost_used_per_second); io.f(m.jump_accel); io.f(m.jump_immediate_force);
        io.f(m.ball_max_speed); io.f(m.ball_damp_per_tick); io.f(m.respawn_delay); io.f(m.bump_cooldown); io.f(m.pad_cooldown_big); io.f(m.pad_cooldown_small);
        io.f(m.spawn_boost); io.f(m.ball_hit_extra_scale); io.f(m.bump_force_scale); io.f(m.goal_threshold_y); io.u(m.flags);
        io.f(m.gravity_x); io.f(m.gravity_y); io.f(m.car_world_friction); io.f(m.car_world_restitution); io.f(m.ball_world_friction); io.f(m.ball_world_restitution);
    }
}

template <int NC>
constexpr size_t arena_num_words() {
    // MUST equal what arena_visit visits: the staged load / store loops of the kernels run over this many word rows of n_envs words each,
    // the allocation is sized by the visitor's own count.  (Until round 4 this said NC * 90 for the 89 words of a car: every step and
    // collection launch read and WROTE NC rows past the end of the resident words -- harmless while the rows fell into the allocation's
    // page slack, garbage in the allocation behind it (the action table) when they did not.)  rlgpu_env_create refuses to run on a
    // mismatch, rlgpu_state_word_counts reports both numbers to the CPU tests.
#ifdef RLG_TEST_EXTRA_WORD_ROWS   /* test build only (tools/build_variant.sh): the old defect on purpose, so the redzone test can be seen to catch it */
    return 4 + 21 + (size_t)NC * 89 + 68 + 10 + (size_t)(NC + 2) / 2 + (size_t)NC * (8 + RLGPU_NUM_EVENT_VALS + 1) + 4 + MUTATOR_WORDS + (size_t)NC * RLG_TEST_EXTRA_WORD_ROWS;
#else
    return 4 + 21 + (size_t)NC * 89 + 68 + 10 + (size_t)(NC + 2) / 2 + (size_t)NC * (8 + RLGPU_NUM_EVENT_VALS + 1) + 4 + MUTATOR_WORDS;
#endif
}

// finish a freshly loaded working copy: derived values that are not stored.  The ball's basis (BallState::rotMat) IS stored (nine resident
// words): every built-in state setter leaves the identity, a user setter's basis arrives through arena_from_host.
template <int NC>
RLG_HD void arena_finish_load(Arena<NC>& A) {
    if (A.ball.b.rot.r0.x == 0.f && A.ball.b.rot.r0.y == 0.f && A.ball.b.rot.r0.z == 0.f) A.ball.b.rot = m3_identity();   // (words that never saw a setter or arena_from_host: all zero)
    A.ball.b.force = v3(0, 0, 0); A.ball.b.torque = v3(0, 0, 0);
    body_update_inertia(A.ball.b, ball_inv_inertia_local());
    for (int k = 0; k < NC; k++) {
        Car& c = A.cars[k];
        c.b.force = v3(0, 0, 0); c.b.torque = v3(0, 0, 0); c.frozen = false;
        if (!(c.flags & CF_IS_DEMOED)) body_update_inertia(c.b, car_inv_inertia_local());    // (demoed: the slot holds car_ghost_rot)
    }
    for (int p = 0; p < 34; p++) A.pads[p].cur_locked = 0;
}

// ---- (1) <-> (2) ----------------------------------------------------------------------------------------
RLG_HD V3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }
RLG_HD void st3(float* p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }
RLG_HD Controls ctl_from(const float* p) {
    Controls c; c.throttle = p[0]; c.steer = p[1]; c.pitch = p[2]; c.yaw = p[3]; c.roll = p[4];
    c.jump = p[5] != 0.f; c.boost = p[6] != 0.f; c.handbrake = p[7] != 0.f; return c;
}
RLG_HD void ctl_to(float* p, const Controls& c) {
    p[0] = c.throttle; p[1] = c.steer; p[2] = c.pitch; p[3] = c.yaw; p[4] = c.roll;
    p[5] = c.jump ? 1.f : 0.f; p[6] = c.boost ? 1.f : 0.f; p[7] = c.handbrake ? 1.f : 0.f;
}

RLG_HD Mutators mutators_from_abi(const RlgpuMutators& a) {
    Mutators m;
    m.gravity_z = a.gravity_z; m.boost_accel_ground = a.boost_accel_ground; m.boost_accel_air = a.boost_accel_air; m.boost_used_per_second = a.boost_used_per_second;
    m.jump_accel = a.jump_accel; m.jump_immediate_force = a.jump_immediate_force; m.ball_max_speed = a.ball_max_speed; m.ball_damp_per_tick = a.ball_damp_per_tick;
    m.respawn_delay = a.respawn_delay; m.bump_cooldown = a.bump_cooldown_time; m.pad_cooldown_big = a.boost_pad_cooldown_big; m.pad_cooldown_small = a.boost_pad_cooldown_small;
    m.spawn_boost = a.car_spawn_boost_amount; m.ball_hit_extra_scale = a.ball_hit_extra_force_scale; m.bump_force_scale = a.bump_force_scale;
    m.goal_threshold_y = a.goal_base_threshold_y; m.flags = a.flags;
    m.gravity_x = a.gravity_x; m.gravity_y = a.gravity_y; m.car_world_friction = a.car_world_friction; m.car_world_restitution = a.car_world_restitution;
    m.ball_world_friction = a.ball_world_friction; m.ball_world_restitution = a.ball_world_restitution;
    return m;
}
RLG_HD RlgpuMutators mutators_to_abi(const Mutators& m) {
    RlgpuMutators a;
    a.gravity_z = m.gravity_z; a.boost_accel_ground = m.boost_accel_ground; a.boost_accel_air = m.boost_accel_air; a.boost_used_per_second = m.boost_used_per_second;
    a.jump_accel = m.jump_accel; a.jump_immediate_force = m.jump_immediate_force; a.ball_max_speed = m.ball_max_speed; a.ball_damp_per_tick = m.ball_damp_per_tick;
    a.respawn_delay = m.respawn_delay; a.bump_cooldown_time = m.bump_cooldown; a.boost_pad_cooldown_big = m.pad_cooldown_big; a.boost_pad_cooldown_small = m.pad_cooldown_small;
    a.car_spawn_boost_amount = m.spawn_boost; a.ball_hit_extra_force_scale = m.ball_hit_extra_scale; a.bump_force_scale = m.bump_force_scale;
    a.goal_base_threshold_y = m.goal_threshold_y; a.flags = m.flags; a._pad = 0u;
    a.gravity_x = m.gravity_x; a.gravity_y = m.gravity_y; a.car_world_friction = m.car_world_friction; a.car_world_restitution = m.car_world_restitution;
    a.ball_world_friction = m.ball_world_friction; a.ball_world_restitution = m.ball_world_restitution;
    return a;
}

template <int NC>
RLG_HD void arena_from_host(Arena<NC>& A, GymEnv<NC>& G, const RlgpuArenaState& s) {
    A.tick_count = s.tick_count; A.ball_update_counter = s.ball_update_counter;
    A.car_order = car_order_checked(s.car_order, NC);
    // the broadphase's memory of its proxies (RlgpuArenaHidden::bp_hist): what a download handed out goes back in; a struct without it (valid bit 0
    // clear: every recording, every state a user builds) means a fresh arena set to this state -- an env slot keeps its own history then (k_upload)
    for (int b = 0; b <= NC; b++) A.bp_hist[b] = (s.hidden.valid & RLGPU_HIDDEN_BP_HIST) ? s.hidden.bp_hist[b] : (uint16_t)0;
    A.ref_engine = (s.hidden.valid & RLGPU_HIDDEN_REF_ENGINE) ? s.hidden.ref_engine : 0u;   // (k_upload keeps the slot's engine when the state brings none)
    if (s.hidden.valid & RLGPU_HIDDEN_MUTATORS) A.mut = mutators_from_abi(s.mutators); else A.mut = mutators_default();   // (k_upload: likewise)
    A.ball.b.pos = ld3(s.ball.pos) * UU2BT; A.ball.b.vel = ld3(s.ball.vel) * UU2BT; A.ball.b.angvel = ld3(s.ball.ang_vel);
    A.ball.vel_impulse_cache = ld3(s.ball.vel_impulse_cache) * UU2BT;
    {   // BallState::rotMat from the appended block (all zeros = a caller that knows nothing of it: a default BallState)
        bool all_zero = true;
        for (int q = 0; q < 9; q++) all_zero = all_zero && s.hidden.ball_rot[q] == 0.f;
        A.ball.b.rot = all_zero ? m3_identity() : m3_cols(ld3(s.hidden.ball_rot), ld3(s.hidden.ball_rot + 3), ld3(s.hidden.ball_rot + 6));
    }
    for (int k = 0; k < NC; k++) {
        const RlgpuCarState& o = s.cars[k]; Car& c = A.cars[k];
        c.b.pos = ld3(o.pos) * UU2BT;
        c.b.rot = m3_cols(ld3(o.rot), ld3(o.rot + 3), ld3(o.rot + 6));
        c.b.vel = ld3(o.vel) * UU2BT; c.b.angvel = ld3(o.ang_vel);
        c.flags = o.flags; c.flip_rel_torque = ld3(o.flip_rel_torque);
        c.jump_time = o.jump_time; c.flip_time = o.flip_time; c.air_time = o.air_time; c.air_time_since_jump = o.air_time_since_jump;
        c.boost = o.boost; c.time_spent_boosting = o.time_spent_boosting; c.supersonic_time = o.supersonic_time;
        c.handbrake_val = o.handbrake_val; c.auto_flip_timer = o.auto_flip_timer; c.auto_flip_torque_scale = o.auto_flip_torque_scale;
        c.world_contact_normal = ld3(o.world_contact_normal); c.car_contact_other = o.car_contact_other_id;
        c.car_contact_cooldown = o.car_contact_cooldown; c.demo_respawn_timer = o.demo_respawn_timer;
        c.bh_rel_pos = ld3(o.bh_rel_pos); c.bh_ball_pos = ld3(o.bh_ball_pos); c.bh_extra_hit_vel = ld3(o.bh_extra_hit_vel);
        c.bh_tick_hit = o.bh_tick_hit; c.bh_tick_extra = o.bh_tick_extra;
        c.last = ctl_from(o.last_controls); c.ctl = ctl_from(o.controls);
        c.vel_impulse_cache = ld3(o.vel_impulse_cache) * UU2BT;
        for (int w = 0; w < 4; w++) { c.extra_pushback[w] = o.extra_pushback[w]; c.lat_friction[w] = o.wheel_lat_friction[w]; c.long_friction[w] = o.wheel_long_friction[w]; }
        c.steer_angle = o.wheel_steer_angle; c.engine_force = o.wheel_engine_force; c.brake = o.wheel_brake;
        c.b.inv_inertia_w = c.b.rot;     // car_ghost_rot of a demolished car: Car::SetState gives the rigid body the reported basis (Car.cpp:22-36) ...
        if ((s.hidden.valid & RLGPU_HIDDEN_WRECK_ROT) && (o.flags & CF_IS_DEMOED))   // ... unless the state carries the basis the wreck's body really has
            c.b.inv_inertia_w = m3_cols(ld3(s.hidden.wreck_rot[k]), ld3(s.hidden.wreck_rot[k] + 3), ld3(s.hidden.wreck_rot[k] + 6));
    }
    for (int p = 0; p < 34; p++) {
        A.pads[p].cooldown = s.pads[p].cooldown; A.pads[p].is_active = s.pads[p].is_active != 0;
        A.pads[p].prev_locked = (int8_t)s.pads[p].prev_locked_car_id; A.pads[p].cur_locked = 0;
    }
    const RlgpuGymState& g = s.gym;
    G.score_line[0] = g.score_line[0]; G.score_line[1] = g.score_line[1]; G.last_touch_car_id = g.last_touch_car_id;
    G.last_tick_count = g.last_tick_count; G.no_touch_steps = g.no_touch_steps; G.shot_cooldown = g.shot_cooldown;
    G.tracker_flags = (g.ball_shot ? 1u : 0u) | (g.ball_shot_goal_team ? 2u : 0u) | (g.ball_scored_last ? 4u : 0u);
    G.last_ball_update_count = g.last_ball_update_count;
    for (int k = 0; k < NC; k++) {
        const RlgpuPlayerGymState& q = g.players[k];
        G.counters[k][0] = q.match_goals; G.counters[k][1] = q.match_saves; G.counters[k][2] = q.match_assists; G.counters[k][3] = q.match_shots;
        G.counters[k][4] = q.match_shot_passes; G.counters[k][5] = q.match_bumps; G.counters[k][6] = q.match_demos; G.counters[k][7] = q.boost_pickups;
        for (int e = 0; e < RLGPU_NUM_EVENT_VALS; e++) G.event_last[k][e] = q.event_last[e];
        G.prev_action_idx[k] = q.prev_action_idx;
    }
    G.tracker_flags |= (g.snap_demoed_mask & 0xffu) << 8;
    G.episode_steps = g.episode_steps; G.reset_count = g.reset_count;
    arena_finish_load(A);
}

template <int NC>
RLG_HD void arena_to_host(const Arena<NC>& A, const GymEnv<NC>& G, RlgpuArenaState& s) {
    s.num_cars = NC; s.car_order = A.car_order;
    s.tick_count = A.tick_count; s.ball_update_counter = A.ball_update_counter;
    st3(s.ball.pos, A.ball.b.pos * BT2UU); st3(s.ball.vel, A.ball.b.vel * BT2UU); st3(s.ball.ang_vel, A.ball.b.angvel);
    st3(s.ball.vel_impulse_cache, A.ball.vel_impulse_cache * BT2UU);
    st3(s.hidden.ball_rot, col0(A.ball.b.rot)); st3(s.hidden.ball_rot + 3, col1(A.ball.b.rot)); st3(s.hidden.ball_rot + 6, col2(A.ball.b.rot));
    // the arena's other hidden state: the broadphase history of the dynamic proxies, and the basis a demolished car's rigid body has turned to
    // behind the stale rotation its state reports (car_ghost_rot, arena_world.h)
    s.hidden.valid = RLGPU_HIDDEN_BP_HIST | RLGPU_HIDDEN_WRECK_ROT | RLGPU_HIDDEN_REF_ENGINE | RLGPU_HIDDEN_MUTATORS;
    s.mutators = mutators_to_abi(A.mut);
    s.hidden.ref_engine = A.ref_engine; s.hidden._pad = 0;
    for (int b = 0; b < 8; b++) s.hidden.bp_hist[b] = b <= NC ? A.bp_hist[b] : (uint16_t)0;
    for (int k = 0; k < RLGPU_MAX_CARS; k++) for (int q = 0; q < 9; q++) s.hidden.wreck_rot[k][q] = 0.f;
    for (int k = 0; k < NC; k++) if (A.cars[k].flags & CF_IS_DEMOED) { const M3& g = A.cars[k].b.inv_inertia_w; st3(s.hidden.wreck_rot[k], col0(g)); st3(s.hidden.wreck_rot[k] + 3, col1(g)); st3(s.hidden.wreck_rot[k] + 6, col2(g)); }
    for (int k = 0; k < NC; k++) {
        RlgpuCarState& o = s.cars[k]; const Car& c = A.cars[k];
        st3(o.pos, c.b.pos * BT2UU);
        st3(o.rot, col0(c.b.rot)); st3(o.rot + 3, col1(c.b.rot)); st3(o.rot + 6, col2(c.b.rot));
        st3(o.vel, c.b.vel * BT2UU); st3(o.ang_vel, c.b.angvel);
        o.flags = c.flags; st3(o.flip_rel_torque, c.flip_rel_torque);
        o.jump_time = c.jump_time; o.flip_time = c.flip_time; o.air_time = c.air_time; o.air_time_since_jump = c.air_time_since_jump;
        o.boost = c.boost; o.time_spent_boosting = c.time_spent_boosting; o.supersonic_time = c.supersonic_time;
        o.handbrake_val = c.handbrake_val; o.auto_flip_timer = c.auto_flip_timer; o.auto_flip_torque_scale = c.auto_flip_torque_scale;
        st3(o.world_contact_normal, c.world_contact_normal); o.car_contact_other_id = c.car_contact_other;
        o.car_contact_cooldown = c.car_contact_cooldown; o.demo_respawn_timer = c.demo_respawn_timer;
        st3(o.bh_rel_pos, c.bh_rel_pos); st3(o.bh_ball_pos, c.bh_ball_pos); st3(o.bh_extra_hit_vel, c.bh_extra_hit_vel);
        o.bh_tick_hit = c.bh_tick_hit; o.bh_tick_extra = c.bh_tick_extra;
        ctl_to(o.last_controls, c.last); ctl_to(o.controls, c.ctl);
        st3(o.vel_impulse_cache, c.vel_impulse_cache * BT2UU);
        for (int w = 0; w < 4; w++) { o.extra_pushback[w] = c.extra_pushback[w]; o.wheel_lat_friction[w] = c.lat_friction[w]; o.wheel_long_friction[w] = c.long_friction[w]; }
        o.wheel_steer_angle = c.steer_angle; o.wheel_engine_force = c.engine_force; o.wheel_brake = c.brake;
    }
    for (int p = 0; p < 34; p++) {
        s.pads[p].cooldown = A.pads[p].cooldown; s.pads[p].is_active = A.pads[p].is_active ? 1 : 0;
        s.pads[p]._pad[0] = s.pads[p]._pad[1] = s.pads[p]._pad[2] = 0;
        s.pads[p].prev_locked_car_id = A.pads[p].prev_locked;
    }
    RlgpuGymState& g = s.gym;
    g.score_line[0] = G.score_line[0]; g.score_line[1] = G.score_line[1]; g.last_touch_car_id = G.last_touch_car_id;
    g.last_tick_count = G.last_tick_count; g.no_touch_steps = G.no_touch_steps; g.shot_cooldown = G.shot_cooldown;
    g.ball_shot = G.tracker_flags & 1u ? 1 : 0; g.ball_shot_goal_team = G.tracker_flags & 2u ? 1 : 0; g.ball_scored_last = G.tracker_flags & 4u ? 1 : 0; g._pad0 = 0;
    g.last_ball_update_count = G.last_ball_update_count;
    for (int k = 0; k < NC; k++) {
        RlgpuPlayerGymState& q = g.players[k];
        q.match_goals = G.counters[k][0]; q.match_saves = G.counters[k][1]; q.match_assists = G.counters[k][2]; q.match_shots = G.counters[k][3];
        q.match_shot_passes = G.counters[k][4]; q.match_bumps = G.counters[k][5]; q.match_demos = G.counters[k][6]; q.boost_pickups = G.counters[k][7];
        for (int e = 0; e < RLGPU_NUM_EVENT_VALS; e++) q.event_last[e] = G.event_last[k][e];
        for (int e = 0; e < 8; e++) q.prev_action[e] = 0.f;  // the table row is filled in by the host side (it owns the table)
        q.prev_action_idx = G.prev_action_idx[k];
    }
    g.snap_demoed_mask = (G.tracker_flags >> 8) & 0xffu; g.episode_steps = G.episode_steps; g.reset_count = G.reset_count; g._pad1 = 0;
}

}  // namespace rlg
