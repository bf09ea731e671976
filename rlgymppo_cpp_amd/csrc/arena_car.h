// arena_car.h — the per-car part of one physics tick: suspension rays, wheel friction, drive/brake/steer
// curves, jump / flip / double-jump, air torque, auto-flip, auto-roll, boost, post-tick bookkeeping.
// Restates RocketSim/src/Sim/Car/Car.cpp:58-193,330-833 and RocketSim/src/Sim/btVehicleRL/btVehicleRL.cpp:64-402
// for a fixed Octane (CarConfig.cpp:20-70).  All vectors are in BT units (uu/50) like the reference's btRigidBody.
#pragma once
#include <cstddef>
#include "arena_body.h"
#include "arena_world.h"

namespace rlg {

// Angle::FromRotMat roll (MathTypes.cpp:72-82 -> btMatrix3x3::getEulerYPR)
RLG_HD float rot_roll(const M3& m) {
    float yaw = rl_atan2f(m.r1.x, m.r0.x);
    float pitch = rl_asinf(-m.r2.x);
    float roll = rl_atan2f(m.r2.y, m.r2.z);
    const float HALF_PI = 1.57079632679489661923f;
    if (fabsf(pitch) == HALF_PI) {
        if (roll > 0) roll -= PI_F; else roll += PI_F;
    }
    (void)yaw;
    return -roll;
}

struct CarTickCtx {
    WheelTmp w[4];
    M3 wheel_basis[2];   // [0] both front wheels (steered), [1] both back wheels
    unsigned long long ray_key[4];   // closest mesh hit of each wheel's ray (arena_world.h:ray_key)
    int n_contact;
    bool wheels_world;
    float forward_speed_uu;
    float new_lat[4], new_long[4];   // this tick's friction factors per wheel (car_wheel_trace), adopted by car_pre_tick_finish
};

// phase 2's private copy of what the wheel lanes left in the car's CarTickCtx (car_pre_tick_finish): read in one go into registers -- the context
// lives in LDS behind a reference, where every conditional read is a dependent round trip and every store in between forces the reads after it
struct CarWheels {
    WheelTmp w[4];
    float new_lat[4], new_long[4];
    int n_contact;
    bool wheels_world;
    float forward_speed_uu;
};

// true when phase 2 of car `ci` reads another car (a wheel stands on it): such ticks run phase 2 in car order
RLG_HD bool car_needs_ordered_finish(const CarTickCtx& t) {
    // (bitwise on purpose: `||` / `&&` over values behind an LDS reference compile to one dependent load + branch per term, a chain of ~100-cycle round trips)
    const bool c0 = t.w[0].in_contact, c1 = t.w[1].in_contact, c2 = t.w[2].in_contact, c3 = t.w[3].in_contact;
    const int g0 = t.w[0].ground, g1 = t.w[1].ground, g2 = t.w[2].ground, g3 = t.w[3].ground;
    return (c0 & (g0 >= 2)) | (c1 & (g1 >= 2)) | (c2 & (g2 >= 2)) | (c3 & (g3 >= 2));
}

// per-wheel part of Car::_UpdateWheels (Car.cpp:405-452): this tick's lateral / longitudinal friction factors of wheel i.
// Needs only state that nothing changes before car_pre_tick_finish, so it runs on the wheel's own lane.
RLG_HD void wheel_friction_factors(const CarHot& c, const WheelTmp& w, const M3& basis, float& latf_out, float& lonf_out) {
    const float dt = TICK_DT;
    float hb = c.handbrake_val;   // the value car_update_wheels is about to store
    if (c.ctl.handbrake) hb += K::POWERSLIDE_RISE_RATE * dt; else hb -= K::POWERSLIDE_FALL_RATE * dt;
    hb = clampf(hb, 0.f, 1.f);
    float real_throttle = c.ctl.throttle;
    if (c.ctl.boost && c.boost > 0) real_throttle = 1.f;
    V3 lat = col1(basis);
    V3 lon = cross(lat, w.contact_normal);
    float input = 0.f;
    V3 wheel_delta = w.hard_point - c.b.pos;
    V3 cv = (cross(c.b.angvel, wheel_delta) + c.b.vel) * BT2UU;
    float base = fabsf(dot(cv, lat));
    if (base > 5.f) input = base / (fabsf(dot(cv, lon)) + base);
    float latf = curve_lat_friction(input);
    float lonf = 1.f;  // LONG_FRICTION_CURVE is empty -> default output 1 (RLConst.h:376-380, Math.cpp:31-33)
    if (hb != 0.f) {
        latf *= (0.1f - 1.f) * hb + 1.f;  // HANDBRAKE_LAT_FRICTION_FACTOR_CURVE is the constant 0.1 (RLConst.h:382-386)
        lonf *= (curve_handbrake_long(input) - 1.f) * hb + 1.f;
    } else {
        lonf = 1.f;
    }
    if (real_throttle == 0.f) {
        float s = curve_non_sticky(w.contact_normal.z);
        latf *= s; lonf *= s;
    }
    latf_out = latf; lonf_out = lonf;
}

// one wheel of calcFrictionImpulses, with LAST tick's engine force / brake / friction factors (btVehicleRL.cpp:313-387)
template <int NC>
RLG_HD V3 wheel_friction_impulse(const Arena<NC>& A, const CarHot& c, const WheelTmp& w, const M3& basis, int i) {
    const float friction_scale = K::CAR_MASS / 3;
    if (w.ground < 0) return v3(0, 0, 0);
    V3 axle = col1(basis);
    V3 n = w.contact_normal;
    float proj = dot(axle, n);
    axle -= n * proj;
    axle = safe_normalized(axle);
    V3 fdir = safe_normalized(cross(n, axle));
    // resolveSingleBilateral (btContactConstraint.cpp:108-155)
    V3 rel1 = w.contact_point - c.b.pos;
    V3 vel1 = body_vel_at(c.b, rel1);
    V3 vel2 = v3(0, 0, 0);
    float g_dot = 0.f;
    const Body* gb = nullptr; float g_inv_mass = 0.f; V3 g_inv_inertia = v3(0, 0, 0);
    if (w.ground == 1) { gb = &A.ball.b; g_inv_mass = BALL_INV_MASS; g_inv_inertia = ball_inv_inertia_local(); }
    else if (w.ground >= 2) { gb = &A.cars[w.ground - 2].b; g_inv_mass = CAR_INV_MASS; g_inv_inertia = car_inv_inertia_local(); }
    V3 rel2 = v3(0, 0, 0);
    if (gb) {
        rel2 = w.contact_point - gb->pos;
        vel2 = body_vel_at(*gb, rel2);
        // (a wreck met on the tick after its demolition, ray_ball_and_cars: the rigid body's basis, not the reported one)
        const M3 grot = (w.ground >= 2 && (A.cars[w.ground - 2].flags & CF_IS_DEMOED)) ? car_ghost_rot(A.cars[w.ground - 2]) : gb->rot;
        V3 bJ = tmul(grot, cross(rel2, -axle));
        g_dot = dot(g_inv_inertia * bJ, bJ);
    }
    V3 aJ = tmul(c.b.rot, cross(rel1, axle));  // world2A * (rel_pos1 x normal), world2A = basis^T
    // btJacobianEntry (btJacobianEntry.h:50): m_Adiag = massInvA + m_0MinvJt.dot(m_aJ) + massInvB + m_1MinvJt.dot(m_bJ), summed left to right
    float diag = ((CAR_INV_MASS + dot(car_inv_inertia_local() * aJ, aJ)) + g_inv_mass) + g_dot;
    float rel_vel = dot(axle, vel1 - vel2);
    float side_impulse = -0.2f * rel_vel * (1.f / diag);
    float rolling;
    if (c.engine_force == 0.f) {
        if (c.brake != 0.f) {
            V3 v2r = gb ? body_vel_at(*gb, rel1) : v3(0, 0, 0);  // the reference uses carRelContactPoint for both (btVehicleRL.cpp:349-352)
            float rv = dot(vel1 - v2r, fdir);
            rolling = clampf(-rv * 113.73963f, -c.brake, c.brake);
        } else rolling = 0.f;
    } else {
        rolling = -c.engine_force / friction_scale;
    }
    V3 total = (fdir * rolling * c.long_friction[i]) + (axle * side_impulse * c.lat_friction[i]);
    return total * friction_scale;
}

// ---- Car::_UpdateWheels (Car.cpp:330-475) -------------------------------------------------------------
RLG_HD void car_update_wheels(CarHot& c, const CarWheels& t) {
    const float dt = TICK_DT;
    float abs_fwd = fabsf(t.forward_speed_uu);
    if (c.ctl.handbrake) c.handbrake_val += K::POWERSLIDE_RISE_RATE * dt;
    else c.handbrake_val -= K::POWERSLIDE_FALL_RATE * dt;
    c.handbrake_val = clampf(c.handbrake_val, 0.f, 1.f);

    float real_throttle = c.ctl.throttle, real_brake = 0.f;
    if (c.ctl.boost && c.boost > 0) real_throttle = 1.f;
    {
        float drive_scale = curve_drive_torque(abs_fwd);
        float engine_throttle = real_throttle;
        if (!c.ctl.handbrake) {
            float abs_thr = fabsf(real_throttle);
            if (abs_thr >= K::THROTTLE_DEADZONE) {
                if (abs_fwd > K::STOPPING_FORWARD_VEL && sgnf(real_throttle) != sgnf(t.forward_speed_uu)) {
                    real_brake = 1.f;
                    if (abs_fwd > K::BRAKING_NO_THROTTLE_SPEED_THRESH) engine_throttle = 0.f;
                }
            } else {
                engine_throttle = 0.f;
                real_brake = (abs_fwd < K::STOPPING_FORWARD_VEL) ? 1.f : K::COASTING_BRAKE_FACTOR;
            }
        }
        if (t.n_contact < 3) drive_scale /= 4.f;
        c.engine_force = engine_throttle * (K::THROTTLE_TORQUE_AMOUNT * UU2BT) * drive_scale;
        c.brake = real_brake * (K::BRAKE_TORQUE_AMOUNT * UU2BT);
    }
    {
        float steer = curve_steer_angle(abs_fwd);
        if (c.handbrake_val != 0.f) steer += (curve_powerslide_steer(abs_fwd) - steer) * c.handbrake_val;
        steer *= c.ctl.steer;
        c.steer_angle = steer;
    }
    for (int i = 0; i < 4; i++) {
        if (t.w[i].ground < 0) continue;
        c.lat_friction[i] = t.new_lat[i];     // computed on the wheel's lane (wheel_friction_factors)
        c.long_friction[i] = t.new_long[i];
    }
    if (t.wheels_world) {
        V3 sum = v3(0, 0, 0);
        for (int i = 0; i < 4; i++) if (t.w[i].in_contact) sum += t.w[i].contact_normal;
        V3 up = is_zero(sum) ? col2(c.b.rot) : safe_normalized(sum);
        bool full_stick = (real_throttle != 0.f) || (abs_fwd > K::STOPPING_FORWARD_VEL);
        float scale = 0.5f;
        if (full_stick) scale += 1.f - fabsf(up.z);
        c.b.force += up * scale * (K::GRAVITY_Z * UU2BT) * K::CAR_MASS;
    }
}

// ---- Car::_UpdateAirTorque (Car.cpp:556-641) ------------------------------------------------------------
RLG_HD void car_update_air_torque(CarHot& c, bool update_air_control) {
    V3 fwd = col0(c.b.rot), right = col1(c.b.rot), up = col2(c.b.rot);
    V3 dir_pitch = -right, dir_yaw = up, dir_roll = -fwd;
    bool do_air = false;
    if (c.flags & CF_IS_FLIPPING) {
        bool keep = (c.flags & CF_HAS_FLIPPED) && c.flip_time < K::FLIP_TORQUE_TIME;
        if (!keep) c.flags &= ~CF_IS_FLIPPING;
    }
    const M3 inertia_w = m3_inverse(c.b.inv_inertia_w);   // the reference inverts m_invInertiaTensorWorld numerically where it needs the world inertia (Car.cpp:590,636,832)
    if (c.flags & CF_IS_FLIPPING) {
        V3 rel = c.flip_rel_torque;
        if (!is_zero(c.flip_rel_torque)) {
            float pitch_scale = 1.f;
            if (rel.y != 0.f && c.ctl.pitch != 0.f) {
                if (sgnf(rel.y) == sgnf(c.ctl.pitch)) {
                    pitch_scale = 1.f - fminf(fabsf(c.ctl.pitch), 1.f);
                    do_air = true;
                }
            }
            rel.y *= pitch_scale;
            V3 dodge = rel * v3(K::FLIP_TORQUE_X, K::FLIP_TORQUE_Y, 0.f);
            c.b.torque += (inertia_w * c.b.rot) * dodge;      // inverse() * basis * dodgeTorque, left to right
        } else {
            do_air = true;
        }
    } else {
        do_air = true;
    }
    do_air = do_air && !(c.flags & CF_IS_AUTOFLIPPING);
    do_air = do_air && update_air_control;
    if (do_air) {
        float pitch_scale = 1.f;
        V3 torque;
        if (c.ctl.pitch != 0.f || c.ctl.yaw != 0.f || c.ctl.roll != 0.f) {
            if (c.flags & CF_IS_FLIPPING) pitch_scale = 0.f;
            else if (c.flags & CF_HAS_FLIPPED) {
                if (c.flip_time < K::FLIP_TORQUE_TIME + K::FLIP_PITCHLOCK_EXTRA_TIME) pitch_scale = 0.f;
            }
            torque = (c.ctl.pitch * dir_pitch * pitch_scale * K::AIR_TORQUE_P) + (c.ctl.yaw * dir_yaw * K::AIR_TORQUE_Y) +
                     (c.ctl.roll * dir_roll * K::AIR_TORQUE_R);
        } else {
            torque = v3(0, 0, 0);
        }
        V3 av = c.b.angvel;
        float damp_pitch = dot(dir_pitch, av) * K::AIR_DAMP_P * (1.f - fabsf(c.ctl.pitch * pitch_scale));
        float damp_yaw = dot(dir_yaw, av) * K::AIR_DAMP_Y * (1.f - fabsf(c.ctl.yaw));
        float damp_roll = dot(dir_roll, av) * K::AIR_DAMP_R;
        V3 damping = (dir_yaw * damp_yaw) + (dir_pitch * damp_pitch) + (dir_roll * damp_roll);
        c.b.torque += (inertia_w * (torque - damping)) * K::CAR_TORQUE_SCALE;
    }
    if (c.ctl.throttle != 0.f) c.b.force += fwd * c.ctl.throttle * K::THROTTLE_AIR_ACCEL * UU2BT * K::CAR_MASS;
}

// ---- Car::_UpdateJump (Car.cpp:507-554) -------------------------------------------------------------------
RLG_HD void car_update_jump(CarHot& c, bool jump_pressed, const Mutators& m) {
    const float dt = TICK_DT;
    bool on_ground = c.flags & CF_ON_GROUND;
    if (on_ground && !(c.flags & CF_IS_JUMPING)) {
        if ((c.flags & CF_HAS_JUMPED) && c.jump_time < K::JUMP_MIN_TIME + K::JUMP_RESET_TIME_PAD) {
        } else {
            c.flags &= ~CF_HAS_JUMPED;
            c.jump_time = 0.f;
        }
    }
    V3 up = col2(c.b.rot);
    if (c.flags & CF_IS_JUMPING) {
        if (c.jump_time < K::JUMP_MIN_TIME || (c.ctl.jump && c.jump_time < K::JUMP_MAX_TIME)) {
        } else {
            c.flags &= ~CF_IS_JUMPING;
        }
    } else if (RLG_UNLIKELY(on_ground && jump_pressed)) {
        c.flags |= CF_IS_JUMPING;
        c.jump_time = 0.f;
        V3 imp = up * m.jump_immediate_force * UU2BT * K::CAR_MASS;   // MutatorConfig::jumpImmediateForce (Car.cpp:532)
        body_apply_central_impulse(c.b, imp, CAR_INV_MASS);
    }
    if (c.flags & CF_IS_JUMPING) {
        c.flags |= CF_HAS_JUMPED;
        V3 f = up * m.jump_accel;   // MutatorConfig::jumpAccel (Car.cpp:540)
        if (c.jump_time < K::JUMP_MIN_TIME) f *= 0.62f;
        c.b.force += f * UU2BT * K::CAR_MASS;
    }
    if (c.flags & (CF_IS_JUMPING | CF_HAS_JUMPED)) c.jump_time += dt;
}

// ---- Car::_UpdateAutoFlip (Car.cpp:763-797) ---------------------------------------------------------------
RLG_HD void car_update_auto_flip(CarHot& c, bool jump_pressed) {
    const float dt = TICK_DT;
    if (RLG_UNLIKELY(jump_pressed && (c.flags & CF_WORLD_CONTACT) && c.world_contact_normal.z > K::CAR_AUTOFLIP_NORMZ_THRESH)) {
        float roll = rot_roll(c.b.rot);
        float abs_roll = fabsf(roll);
        if (abs_roll > K::CAR_AUTOFLIP_ROLL_THRESH) {
            c.auto_flip_timer = K::CAR_AUTOFLIP_TIME * (abs_roll / PI_F);
            c.auto_flip_torque_scale = (roll > 0) ? 1.f : -1.f;
            c.flags |= CF_IS_AUTOFLIPPING;
            body_apply_central_impulse(c.b, -col2(c.b.rot) * K::CAR_AUTOFLIP_IMPULSE * UU2BT * K::CAR_MASS, CAR_INV_MASS);
        }
    }
    if (RLG_UNLIKELY(c.flags & CF_IS_AUTOFLIPPING)) {
        if (c.auto_flip_timer <= 0.f) {
            c.flags &= ~CF_IS_AUTOFLIPPING;
            c.auto_flip_timer = 0.f;
        } else {
            c.b.angvel += col0(c.b.rot) * K::CAR_AUTOFLIP_TORQUE * c.auto_flip_torque_scale * dt;
            c.auto_flip_timer -= dt;
        }
    }
}

// ---- Car::_UpdateDoubleJumpOrFlip (Car.cpp:643-761) -------------------------------------------------------
RLG_HD void car_update_double_jump_or_flip(CarHot& c, bool jump_pressed, float forward_speed_uu, const Mutators& m) {
    const float dt = TICK_DT;
    if (c.flags & CF_ON_GROUND) {
        c.flags &= ~(CF_HAS_DOUBLE_JUMPED | CF_HAS_FLIPPED);
        c.air_time = 0.f; c.air_time_since_jump = 0.f; c.flip_time = 0.f;
    } else {
        c.air_time += dt;
        if ((c.flags & CF_HAS_JUMPED) && !(c.flags & CF_IS_JUMPING)) c.air_time_since_jump += dt;
        else c.air_time_since_jump = 0.f;

        if (RLG_UNLIKELY(jump_pressed && c.air_time_since_jump < K::DOUBLEJUMP_MAX_DELAY)) {
            float mag = fabsf(c.ctl.yaw) + fabsf(c.ctl.pitch) + fabsf(c.ctl.roll);
            bool is_flip = mag >= K::DODGE_DEADZONE;
            bool can_use = (!(c.flags & CF_HAS_DOUBLE_JUMPED) && !(c.flags & CF_HAS_FLIPPED)) || (m.flags & (is_flip ? MUT_UNLIMITED_FLIPS : MUT_UNLIMITED_DOUBLE_JUMPS));   // Car.cpp:665-671
            if (c.flags & CF_IS_AUTOFLIPPING) can_use = false;
            if (can_use) {
                if (is_flip) {
                    c.flip_time = 0.f;
                    c.flags |= CF_HAS_FLIPPED | CF_IS_FLIPPING;
                    float ratio = fabsf(forward_speed_uu) / K::CAR_MAX_SPEED;
                    V3 dodge = v3(-c.ctl.pitch, c.ctl.yaw + c.ctl.roll, 0.f);
                    if (fabsf(c.ctl.yaw + c.ctl.roll) < 0.1f && fabsf(c.ctl.pitch) < 0.1f) dodge = v3(0, 0, 0);
                    else dodge = safe_normalized(dodge);
                    c.flip_rel_torque = v3(-dodge.y, dodge.x, 0.f);
                    if (fabsf(dodge.x) < 0.1f) dodge.x = 0.f;
                    if (fabsf(dodge.y) < 0.1f) dodge.y = 0.f;
                    bool fuzzy_zero = len2(dodge) < SIMD_EPS * SIMD_EPS;  // btVector3::fuzzyZero
                    if (!fuzzy_zero) {
                        bool backwards;
                        if (fabsf(forward_speed_uu) < 100.0f) backwards = dodge.x < 0.0f;
                        else backwards = (dodge.x >= 0.0f) != (forward_speed_uu >= 0.0f);
                        V3 iv = dodge * K::FLIP_INITIAL_VEL_SCALE;
                        float max_x = backwards ? K::FLIP_BACKWARD_IMPULSE_MAX_SPEED_SCALE : K::FLIP_FORWARD_IMPULSE_MAX_SPEED_SCALE;
                        iv.x *= ((max_x - 1) * ratio) + 1.f;
                        iv.y *= ((K::FLIP_SIDE_IMPULSE_MAX_SPEED_SCALE - 1) * ratio) + 1.f;
                        if (backwards) iv.x *= K::FLIP_BACKWARD_IMPULSE_SCALE_X;
                        V3 f = col0(c.b.rot);
                        float ang = rl_atan2f(f.y, f.x);
                        V3 xdir = v3(rl_cosf(ang), -rl_sinf(ang), 0.f), ydir = v3(rl_sinf(ang), rl_cosf(ang), 0.f);
                        V3 dv = v3(dot(iv, xdir), dot(iv, ydir), 0.f);
                        body_apply_central_impulse(c.b, dv * UU2BT * K::CAR_MASS, CAR_INV_MASS);
                    }
                } else {
                    V3 imp = col2(c.b.rot) * K::JUMP_IMMEDIATE_FORCE * UU2BT * K::CAR_MASS;   // (the double jump keeps RLConst's force: Car.cpp:740 does not read the mutator)
                    body_apply_central_impulse(c.b, imp, CAR_INV_MASS);
                    c.flags |= CF_HAS_DOUBLE_JUMPED;
                }
            }
        }
    }
    if (c.flags & CF_IS_FLIPPING) {
        c.flip_time += dt;
        if (c.flip_time <= K::FLIP_TORQUE_TIME) {
            if (c.flip_time >= K::FLIP_Z_DAMP_START && (c.b.vel.z < 0 || c.flip_time < K::FLIP_Z_DAMP_END))
                c.b.vel.z *= K::FLIP_Z_DAMP_PER_TICK;   // pow(1 - FLIP_Z_DAMP_120, dt / (1 / 120)) with dt = 1 / 120: the base itself
        }
    } else if (c.flags & CF_HAS_FLIPPED) {
        c.flip_time += dt;
    }
}

// ---- Car::_UpdateAutoRoll (Car.cpp:799-833) --------------------------------------------------------------
RLG_HD void car_update_auto_roll(CarHot& c, const CarWheels& t) {
    V3 ground_up;
    if (t.n_contact > 0) {
        V3 sum = v3(0, 0, 0);
        for (int i = 0; i < 4; i++) if (t.w[i].in_contact) sum += t.w[i].contact_normal;
        ground_up = is_zero(sum) ? col2(c.b.rot) : safe_normalized(sum);
    } else {
        ground_up = c.world_contact_normal;
    }
    V3 ground_down = -ground_up;
    V3 fwd = col0(c.b.rot), right = col1(c.b.rot);
    V3 cross_right = cross(ground_up, fwd), cross_fwd = cross(ground_down, cross_right);
    float right_f = 1.f - clampf(dot(right, cross_right), 0.f, 1.f);
    float fwd_f = 1.f - clampf(dot(fwd, cross_fwd), 0.f, 1.f);
    V3 tdir_right = fwd * (dot(right, ground_up) >= 0 ? -1.f : 1.f);
    V3 tdir_fwd = right * (dot(fwd, ground_up) >= 0 ? 1.f : -1.f);
    V3 t_right = tdir_right * right_f, t_fwd = tdir_fwd * fwd_f;
    c.b.force += ground_down * K::CAR_AUTOROLL_FORCE * UU2BT * K::CAR_MASS;
    c.b.torque += (m3_inverse(c.b.inv_inertia_w) * (t_fwd + t_right)) * K::CAR_AUTOROLL_TORQUE;
}

// ---- Car::_UpdateBoost (Car.cpp:477-505) -------------------------------------------------------------------
RLG_HD void car_update_boost(CarHot& c, const Mutators& m) {
    const float dt = TICK_DT;
    if (c.time_spent_boosting > 0) {
        if (!c.ctl.boost && c.time_spent_boosting >= K::BOOST_MIN_TIME) c.time_spent_boosting = 0.f;
        else c.time_spent_boosting += dt;
    } else if (c.ctl.boost) {
        c.time_spent_boosting = dt;
    }
    if (c.boost > 0 && c.time_spent_boosting > 0) {
        c.boost = fmaxf(c.boost - m.boost_used_per_second * dt, 0.f);   // MutatorConfig::boostUsedPerSecond, boostAccelGround / Air (Car.cpp:497-499)
        float acc = (c.flags & CF_ON_GROUND) ? m.boost_accel_ground : m.boost_accel_air;
        c.b.force += (acc * UU2BT) * col0(c.b.rot) * K::CAR_MASS;
    }
    c.boost = fminf(c.boost, K::BOOST_MAX);
}

RLG_HD M3 euler_to_rot(float yaw, float pitch, float roll) {  // Angle::ToRotMat = setEulerYPR(yaw,-pitch,-roll) (MathTypes.cpp:84-89)
    float ez = yaw, ey = -pitch, ex = -roll;
    float ci = rl_cosf(ex), cj = rl_cosf(ey), ch = rl_cosf(ez), si = rl_sinf(ex), sj = rl_sinf(ey), sh = rl_sinf(ez);
    float cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
    return m3_rows(v3(cj * ch, sj * sc - cs, sj * cc + ss), v3(cj * sh, sj * ss + cc, sj * cs - sc), v3(-sj, cj * si, cj * ci));
}

// ---- Car::Respawn (Car.cpp:43-56): spawn slot from the caller's RNG draw ------------------------------------
RLG_HD_COLD void car_respawn(Car& c, bool is_blue, uint32_t rnd, float spawn_boost) {
    const float RX[4] = {-2304, -2688, 2304, 2688};
    int idx = (int)(rnd % 4u);
    Car n = {};
    const float yaw = is_blue ? (PI_F / 2) : (float)((double)(PI_F / 2) + 3.14159265358979323846);   // Angle(spawnPos.yawAng + (blue ? 0 : M_PI), 0, 0): the sum is a double, rounded once (Car.cpp:52)
    n.b.pos = v3(RX[idx], -4608.f * (is_blue ? 1.f : -1.f), K::CAR_RESPAWN_Z) * UU2BT;
    n.b.rot = euler_to_rot(yaw, 0.f, 0.f);   // (Angle::ToRotMat with pitch = roll = 0: the same numbers as the yaw-only matrix up to the SIGN of its zeros, which the rigid body shows for one tick)
    n.b.vel = v3(0, 0, 0); n.b.angvel = v3(0, 0, 0);
    n.flags = CF_ON_GROUND;
    n.boost = spawn_boost;   // MutatorConfig::carSpawnBoostAmount (Car.cpp:72)
    n.bh_tick_hit = -1; n.bh_tick_extra = -1;
    // carried wheel values and controls survive SetState in the reference
    n.ctl = c.ctl;
    n.steer_angle = c.steer_angle; n.engine_force = c.engine_force; n.brake = c.brake;
    for (int i = 0; i < 4; i++) { n.extra_pushback[i] = c.extra_pushback[i]; n.lat_friction[i] = c.lat_friction[i]; n.long_friction[i] = c.long_friction[i]; }
    body_update_inertia(n.b, car_inv_inertia_local());
    c = n;
}

// ---- Car::_PreTickUpdate (Car.cpp:58-131) incl. btVehicleRL first/second halves ------------------------------
// The pre-tick is cut into three phases so that a wavefront can run them over different lane sets (rlgpu_env.hip:
// one lane per car, then one lane per wheel, then one lane per car again); the host build runs them in plain loops.
// Results do not depend on the order of cars inside a phase:
//   * a phase-0 respawn changes only the car itself, and a car that is demoed or was respawned this tick stays
//     `frozen` (DISABLE_SIMULATION + CF_NO_CONTACT_RESPONSE, Car.cpp:69-80) for the whole tick: a ray that meets it first is a miss, whatever its pose details (arena_world.h ray_ball_and_cars);
//   * phase 2 reads another car only when a wheel stands on it (ground >= 2): callers serialise that case by car index.

// phase 0, per car: ClampFix, demo timer and respawn (Car.cpp:60-87).  Returns true when the car's respawn is due and has to draw from the reference's
// engine (Arena::ref_engine, parity tests): those draws go in the arena's car order, so they are left to cars_respawn_ref_engine (the car stays a
// wreck with a timer of zero until then -- a state no other path leaves behind).
template <int NC>
RLG_HD_SMALL bool car_tick_begin(Arena<NC>& A, int ci, uint32_t seed, uint32_t env_id) {
    RLG_ASSUME_LDS(A);
    const float dt = TICK_DT;
    Car& c = A.cars[ci];
    c.ctl.throttle = clampf(c.ctl.throttle, -1.f, 1.f); c.ctl.steer = clampf(c.ctl.steer, -1.f, 1.f);
    c.ctl.pitch = clampf(c.ctl.pitch, -1.f, 1.f); c.ctl.yaw = clampf(c.ctl.yaw, -1.f, 1.f); c.ctl.roll = clampf(c.ctl.roll, -1.f, 1.f);
    bool demoed = (c.flags & CF_IS_DEMOED) != 0;
    c.frozen = demoed;  // rigid body disabled for this tick (Car.cpp:69-80)
    if (RLG_UNLIKELY(demoed)) {
        float tm = fmaxf(c.demo_respawn_timer - dt, 0.f);
        c.demo_respawn_timer = tm;
        if (tm == 0.f) {
            if (A.ref_engine != 0u) return true;
            uint32_t rnd[4];
            philox4(seed, 0x51ED270Bu, env_id, (uint32_t)A.tick_count, 0x100u + (uint32_t)ci, rnd);
            Car n = c;
            car_respawn(n, (ci % 2) == 0, rnd[0], A.mut.spawn_boost);
            n.frozen = true;
            c = n;
        }
    }
    return false;
}
// ... the respawns that draw from the reference's engine: Car::Respawn's Math::RandInt(0, 4) (Car.cpp:48, Math.cpp:44-52) in the order in which
// Arena::Step visits the cars (Arena.cpp:716-812)
template <int NC>
RLG_HD_COLD void cars_respawn_ref_engine(Arena<NC>& A) {
    RefEngine e; e.x = A.ref_engine;
    for (int k = 0; k < NC; k++) {
        const int ci = car_at_rank(A, k);
        Car& c = A.cars[ci];
        if (!(c.flags & CF_IS_DEMOED) || c.demo_respawn_timer != 0.f) continue;
        Car n = c;
        car_respawn(n, (ci % 2) == 0, (uint32_t)e.rand_int(0, 4), A.mut.spawn_boost);
        n.frozen = true;
        c = n;
    }
    A.ref_engine = e.x;
}

// phase 1 = the suspension rays, in three steps so that the mesh part can run as one lane per (ray, candidate) pair:
//   1a per (car, wheel)         car_wheel_ray_begin   wheel transform with LAST tick's steer angle (btVehicleRL.cpp:64-92,218-235),
//                                                      the ray, its hit against the four planes
//   1b per (car, wheel, cand)   car_ray_pair          ray vs one candidate triangle of the car (arena_world.h), closest wins
//   1c per (car, wheel)         car_wheel_ray_finish  ball / other cars, suspension (btVehicleRL.cpp:118-212), hard-stop pushback
//                                                      (btContactConstraint.cpp:60-105), the wheel's friction work
// Between 1a and 1c the ray lives in the wheel's scratch: hard_point = origin, contact_point = end, contact_normal / susp_len /
// ground = plane hit (normal / fraction / kind); the pair winners sit in ray_key[].
RLG_HD unsigned long long* ray_keys(CarTickCtx& t) { return t.ray_key; }

template <int NC>
RLG_HD_MID void car_wheel_ray_begin(Arena<NC>& A, int ci, int i, CarTickCtx& t) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS(t);
    Car& cr = A.cars[ci];
    // NB: as in the reference, a car respawned in phase 0 runs the rest of the pre-tick (Respawn clears the flag,
    // Car.cpp:86-87 checks the NEW state).
    if (RLG_UNLIKELY(cr.flags & CF_IS_DEMOED)) return;
    const M3 rot = cr.b.rot; const V3 pos = cr.b.pos;
    const float steer_angle = cr.steer_angle;
    V3 wheel_dir = rot * v3(0, 0, -1), axle = rot * v3(0, -1, 0);
    {
        V3 wup = -wheel_dir;
        V3 fwd = normalized(cross(wup, axle));
        M3 basis2 = m3_cols(fwd, -axle, wup);
        if (i < 2 && steer_angle != 0.f) {
            M3 steer = quat_to_m3(quat_axis_angle(wup, steer_angle));
            t.wheel_basis[i >> 1] = steer * basis2;
        } else {
            t.wheel_basis[i >> 1] = basis2;  // a zero steering angle gives the exact identity quaternion (0,0,0,1)
        }
    }
    RLG_SPROF(47);
    WheelTmp& w = t.w[i];
    V3 source = (rot * wheel_conn(i)) + pos, target = source + (wheel_dir * wheel_ray_len(i));
    RayHit hit = ray_planes(source, target);
    w.hard_point = source; w.contact_point = target;
    w.contact_normal = hit.normal; w.susp_len = hit.frac; w.ground = hit.kind;
    ray_keys(t)[i] = RAY_NO_HIT;
}

// pair `pair` of car ci: candidate slot = first + pair / 4, wheel = pair % 4 (the 4 wheels of a car share each triangle fetch)
template <int NC>
RLG_HD void car_ray_pair(const Arena<NC>& A, MeshView mesh, const CollideQueue<NC>& Q, int ci, int pair, CarTickCtx& t) {
    const int slot = CollideQueue<NC>::region(1 + ci) + (pair >> 2), i = pair & 3;
    const uint32_t c = queue_cand(Q, slot);
    if (c == CAND_HOLE) return;
    const WheelTmp& w = t.w[i];
    float d;
    if (ray_triangle_pair(mesh.bp, mesh.tris[unpack_cand(c).ref], w.hard_point, w.contact_point, w.susp_len, d)) ray_key_min(ray_keys(t)[i], ray_key(d, slot));
}
template <int NC>
RLG_HD int car_ray_pairs(const Arena<NC>& A, const CollideQueue<NC>& Q, int ci) {
    if (A.cars[ci].flags & CF_IS_DEMOED) return 0;
    return 4 * (int)Q.cand_count[1 + ci];
}

// updateVehicleSecond for one wheel (btVehicleRL.cpp:277-310 suspension, :390-402 friction), up to the products: the velocity changes its two
// impulses make.  Nothing here reads the car's velocities, so the wheel's lane can do it ahead of the car's control phase.
RLG_HD float wheel_suspension_force(const WheelTmp& w, int i) {
    if (!w.in_contact) return 0.f;
    float force = (wheel_rest(i) - w.susp_len) * K::SUSPENSION_STIFFNESS * w.clipped_inv;
    float damp = (w.susp_rel_vel < 0) ? K::WHEELS_DAMPING_COMPRESSION : K::WHEELS_DAMPING_RELAXATION;
    float f = force - (damp * w.susp_rel_vel);
    f *= (i < 2) ? K::SUSPENSION_FORCE_SCALE_FRONT : K::SUSPENSION_FORCE_SCALE_BACK;
    if (f < 0) f = 0;
    return f;
}
RLG_HD void wheel_velocity_deltas(WheelTmp& w, const Body& b, int i, float extra_pushback) {
    const float dt = TICK_DT;
    const float f = wheel_suspension_force(w, i);
    w.susp_nz = f != 0.f;
    if (w.susp_nz) {
        V3 off = w.contact_point - b.pos;
        float scale = (f * dt) + extra_pushback;
        V3 imp = w.contact_normal * scale;
        w.susp_dv = imp * CAR_INV_MASS; w.susp_dw = b.inv_inertia_w * cross(off, imp);
    }
    w.fric_nz = !is_zero(w.impulse);
    if (w.fric_nz) {
        V3 updir = col2(b.rot);
        V3 off = w.contact_point - b.pos;
        float updot = dot(updir, off);
        V3 rel = off - updir * updot;
        V3 imp = w.impulse * dt;
        w.fric_dv = imp * CAR_INV_MASS; w.fric_dw = b.inv_inertia_w * cross(rel, imp);
    }
}

template <int NC>
RLG_HD_MID void car_wheel_ray_finish(Arena<NC>& A, int ci, int i, MeshView mesh, const CollideQueue<NC>& Q, CarTickCtx& t) {
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS(t); RLG_ASSUME_LDS(Q);
    const float dt = TICK_DT;
    Car& cr = A.cars[ci];
    if (RLG_UNLIKELY(cr.flags & CF_IS_DEMOED)) return;
    const M3 rot = cr.b.rot; const V3 pos = cr.b.pos, vel = cr.b.vel, angvel = cr.b.angvel;
    V3 up = col2(rot);
    V3 wheel_dir = rot * v3(0, 0, -1);
    WheelTmp w = t.w[i];
    float rest = wheel_rest(i), radius = wheel_radius(i), travel = wheel_travel();
    V3 source = w.hard_point, target = w.contact_point;
    RayHit hit; hit.kind = w.ground; hit.frac = w.susp_len; hit.normal = w.contact_normal;
    RLG_PROF(0);
    if (RLG_UNLIKELY(Q.overflow)) ray_mesh_walk(mesh, source, target, hit);
    else ray_apply_mesh_key(mesh, Q, ray_keys(t)[i], source, target, hit);
    ray_ball_and_cars(A, ci, source, target, hit);
    RLG_PROF(7); RLG_SPROF(40);
    w.impulse = v3(0, 0, 0);
    w.ground = -1; w.in_contact = false;
    if (hit.kind >= 0) {
        float rt = hit.frac, s = 1.f - rt;
        w.contact_point = v3(s * source.x + rt * target.x, s * source.y + rt * target.y, s * source.z + rt * target.z);
        w.contact_normal = normalized(hit.normal);   // btDefaultVehicleRaycaster::castRay normalises the reported normal once more (btDefaultVehicleRaycaster.cpp:45)
        w.in_contact = true;
        w.ground = hit.kind;
        bool is_static = hit.kind == 0;
        float trace_len = dot(w.hard_point - w.contact_point, up);
        w.susp_len = clampf(trace_len - radius, rest - travel, rest + travel);
        float denom = dot(w.contact_normal, up);
        V3 relpos = w.contact_point - pos;
        V3 vel_at = vel + cross(angvel, relpos);
        float proj_vel = dot(w.contact_normal, vel_at);
        if (denom > 0.1f) {
            float inv = 1.f / denom;
            w.susp_rel_vel = proj_vel * inv; w.clipped_inv = inv;
        } else { w.susp_rel_vel = 0.f; w.clipped_inv = 10.f; }
        if (is_static) {
            float thresh = (rest + radius) - K::SUSPENSION_SUBTRACTION;
            if (trace_len < thresh) {
                // resolveSingleCollision(..., applyImpulses=false) vs a static body (btContactConstraint.cpp:60-105)
                float delta = trace_len - thresh;
                float rel_vel = dot(w.contact_normal, vel_at);
                float pos_err = K::ERP * -delta / dt;
                float vel_err = -(1.0f + 0.f) * rel_vel;
                float denom0 = body_impulse_denom(cr.b, w.contact_point, w.contact_normal, CAR_INV_MASS);
                float jac = 1.f / (denom0 + 0.f);
                float imp = pos_err * jac + vel_err * jac;
                imp = 0.f > imp ? 0.f : imp;
                cr.extra_pushback[i] = imp / 4;
            }
        }
    } else {
        w.contact_point = target;
        w.susp_len = rest + travel; w.susp_rel_vel = 0.f; w.contact_normal = -wheel_dir; w.clipped_inv = 1.f;
        cr.extra_pushback[i] = 0.f;
    }
    RLG_SPROF(41);
    // per-wheel halves of calcFrictionImpulses and _UpdateWheels (see the two helpers above)
    w.impulse = wheel_friction_impulse(A, cr, w, t.wheel_basis[i >> 1], i);
    RLG_SPROF(42);
    if (w.ground >= 0) wheel_friction_factors(cr, w, t.wheel_basis[i >> 1], t.new_lat[i], t.new_long[i]);
    wheel_velocity_deltas(w, cr.b, i, cr.extra_pushback[i]);
    t.w[i] = w;
}

template <int NC>
RLG_HD void car_pre_tick_finish_body(Arena<NC>& A, CarHot& c, int ci, CarTickCtx& t, bool last_jump);

// phase 2, per car: friction impulses, the car's control logic, suspension forces (rest of Car::_PreTickUpdate)
template <int NC>
RLG_HD_BIG void car_pre_tick_finish(Arena<NC>& A, int ci, CarTickCtx& t) {
    // Work on a private copy: A lives in LDS behind a pointer the optimiser must assume aliases everything (every field
    // would be re-loaded after every store); a local Car is promoted to registers.
    RLG_ASSUME_LDS(A); RLG_ASSUME_LDS(t);
    if (A.cars[ci].flags & CF_IS_DEMOED) return;
    CarHot c = A.cars[ci];   // (only the part of the car this phase touches: arena_types.h CarHot)
    car_pre_tick_finish_body(A, c, ci, t, A.cars[ci].last.jump);
    static_cast<CarHot&>(A.cars[ci]) = c;
}

template <int NC>
RLG_HD void car_pre_tick_finish_body(Arena<NC>& A, CarHot& c, int ci, CarTickCtx& ctx, bool last_jump) {
    const float dt = TICK_DT;
    CarWheels t;
    for (int i = 0; i < 4; i++) { t.w[i] = ctx.w[i]; t.new_lat[i] = ctx.new_lat[i]; t.new_long[i] = ctx.new_long[i]; }
    t.n_contact = 0; t.wheels_world = false;
    bool ordered = false;
    for (int i = 0; i < 4; i++) {
        t.n_contact += t.w[i].in_contact ? 1 : 0;
        t.wheels_world |= t.w[i].in_contact & (t.w[i].ground == 0);
        ordered |= t.w[i].in_contact & (t.w[i].ground >= 2);
    }
    // friction impulses: taken from the wheel lanes (car_wheel_trace) unless a wheel stands on another car -- that reads the
    // other car's velocity, which its own phase 2 may already have changed (callers run such ticks in car order)
    if (RLG_UNLIKELY(ordered)) {
        for (int i = 0; i < 4; i++) { t.w[i].impulse = wheel_friction_impulse(A, c, t.w[i], ctx.wheel_basis[i >> 1], i); ctx.w[i].impulse = t.w[i].impulse; }
    }

    bool jump_pressed = c.ctl.jump && !last_jump;
    uint32_t wf = 0;
    for (int i = 0; i < 4; i++) wf |= t.w[i].in_contact ? (CF_WHEEL0 << i) : 0u;
    c.flags = (c.flags & ~(CF_WHEEL0 * 15u)) | wf;
    if (t.n_contact >= 3) c.flags |= CF_ON_GROUND; else c.flags &= ~CF_ON_GROUND;

    t.forward_speed_uu = dot(c.b.vel, col0(c.b.rot)) * BT2UU;
    ctx.n_contact = t.n_contact; ctx.wheels_world = t.wheels_world; ctx.forward_speed_uu = t.forward_speed_uu;   // (the host build's debug dumps read them)
    RLG_SPROF(32);
    car_update_wheels(c, t);
    RLG_SPROF(33);
    if (t.n_contact < 3) car_update_air_torque(c, t.n_contact == 0);
    else c.flags &= ~CF_IS_FLIPPING;
    car_update_jump(c, jump_pressed, A.mut);
    car_update_auto_flip(c, jump_pressed);
    car_update_double_jump_or_flip(c, jump_pressed, t.forward_speed_uu, A.mut);
    if (RLG_UNLIKELY(c.ctl.throttle != 0.f && ((t.n_contact > 0 && t.n_contact < 4) || (c.flags & CF_WORLD_CONTACT)))) car_update_auto_roll(c, t);
    c.flags &= ~CF_WORLD_CONTACT;
    RLG_SPROF(34);

    // updateVehicleSecond: suspension (btVehicleRL.cpp:277-310) then friction impulses (:390-402): the velocity changes were formed by the wheels'
    // lanes (wheel_velocity_deltas); here they are added up, in the reference's order
    if (RLG_LIKELY(!ordered)) {
        for (int i = 0; i < 4; i++) if (t.w[i].susp_nz) { c.b.vel += t.w[i].susp_dv; c.b.angvel += t.w[i].susp_dw; }
        RLG_SPROF(35);
        for (int i = 0; i < 4; i++) if (t.w[i].fric_nz) { c.b.vel += t.w[i].fric_dv; c.b.angvel += t.w[i].fric_dw; }
    } else {   // (a wheel on another car: its friction impulse was formed just now, in car order)
        for (int i = 0; i < 4; i++) {
            const float f = wheel_suspension_force(t.w[i], i);
            if (f != 0.f) {
                const WheelTmp& w = t.w[i];
                V3 off = w.contact_point - c.b.pos;
                float scale = (f * dt) + c.extra_pushback[i];
                body_apply_impulse(c.b, w.contact_normal * scale, off, CAR_INV_MASS);
            }
        }
        RLG_SPROF(35);
        V3 updir = col2(c.b.rot);
        for (int i = 0; i < 4; i++) {
            const WheelTmp& w = t.w[i];
            if (!is_zero(w.impulse)) {
                V3 off = w.contact_point - c.b.pos;
                float updot = dot(updir, off);
                V3 rel = off - updir * updot;
                body_apply_impulse(c.b, w.impulse * dt, rel, CAR_INV_MASS);
            }
        }
    }
    car_update_boost(c, A.mut);
    RLG_SPROF(36);
}

// ---- Car::_PostTickUpdate + _FinishPhysicsTick (Car.cpp:133-193) ------------------------------------------
RLG_HD void car_post_tick(Car& c) {
    const float dt = TICK_DT;
    if (RLG_UNLIKELY(c.flags & CF_IS_DEMOED)) return;
    {
        V3 vuu = c.b.vel * BT2UU;
        float sp2 = len2(vuu);
        bool ss;
        if ((c.flags & CF_IS_SUPERSONIC) && c.supersonic_time < K::SUPERSONIC_MAINTAIN_MAX_TIME)
            ss = sp2 >= K::SUPERSONIC_MAINTAIN_MIN_SPEED * K::SUPERSONIC_MAINTAIN_MIN_SPEED;
        else
            ss = sp2 >= K::SUPERSONIC_START_SPEED * K::SUPERSONIC_START_SPEED;
        if (ss) { c.flags |= CF_IS_SUPERSONIC; c.supersonic_time += dt; }
        else { c.flags &= ~CF_IS_SUPERSONIC; c.supersonic_time = 0.f; }
    }
    if (RLG_UNLIKELY(c.car_contact_cooldown > 0)) c.car_contact_cooldown = fmaxf(c.car_contact_cooldown - dt, 0.f);
    c.last = c.ctl;
    // _FinishPhysicsTick
    if (RLG_UNLIKELY(!is_zero(c.vel_impulse_cache))) { c.b.vel += c.vel_impulse_cache; c.vel_impulse_cache = v3(0, 0, 0); }
    const float vmax = K::CAR_MAX_SPEED * UU2BT;
    if (len2(c.b.vel) > vmax * vmax) c.b.vel = normalized(c.b.vel) * vmax;
    if (RLG_UNLIKELY(len2(c.b.angvel) > K::CAR_MAX_ANG_SPEED * K::CAR_MAX_ANG_SPEED)) c.b.angvel = normalized(c.b.angvel) * K::CAR_MAX_ANG_SPEED;
}

}  // namespace rlg
