/*
 * rlgpu_state.h — host-side (AoS) exchange layout for one arena ("env").
 *
 * This is the plain-C view of everything the batched stepper keeps per env.  It is what
 * rlgpu_upload_states()/rlgpu_download_states() move across the C-ABI (the device keeps the same
 * fields transposed into SoA [field][env] arrays), what the host-side
 * StateSetter / user-plugin fallback reads and writes, and what the parity tests exchange with the
 * oracle.  Field meaning follows the reference types they replace:
 *
 *   RlgpuCarState   <- RocketSim::CarState   (RocketSim/src/Sim/Car/Car.h:17-123)
 *                      + Car::_velocityImpulseCache (Car.h:170), Car::controls (Car.h:147),
 *                      + btWheelInfoRL::m_extraPushback (btVehicleRL.h:24; stale-persistent)
 *   RlgpuBallState  <- RocketSim::BallState  (RocketSim/src/Sim/Ball/Ball.h:17-44) + _velocityImpulseCache
 *   RlgpuPadState   <- RocketSim::BoostPadState (RocketSim/src/Sim/BoostPad/BoostPad.h)
 *   RlgpuGymState   <- RLGSC::Gym::prevState counters, GameEventTracker, terminal-condition and
 *                      EventReward carried state (RLGymSim_CPP/src/RLGymSim_CPP/Gym.h:9-20,
 *                      RocketSim/src/Sim/GameEventTracker/GameEventTracker.h:78-86,
 *                      Utils/TerminalConditions/NoTouchCondition.h:8, Utils/RewardFunctions/CommonRewards.h:15)
 *
 * Units are RocketSim's public units: positions/velocities in uu, angular velocity in rad/s.
 * Rotation is three column vectors forward/right/up (RotMat, MathTypes.h:162).
 * Car slot order inside an env is fixed: slot 2k = blue k, slot 2k+1 = orange k  (the reference's
 * Gym ctor AddCar order, Gym.cpp:45-49; car id = slot + 1).
 */
#ifndef RLGPU_STATE_H
#define RLGPU_STATE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RLGPU_MAX_CARS 6
#define RLGPU_NUM_PADS 34
#define RLGPU_NUM_EVENT_VALS 11

/* flag bits of RlgpuCarState.flags */
enum {
    RLGPU_CF_ON_GROUND       = 1u << 0,
    RLGPU_CF_WHEEL0          = 1u << 1, /* wheelsWithContact[0..3] = bits 1..4 */
    RLGPU_CF_WHEEL1          = 1u << 2,
    RLGPU_CF_WHEEL2          = 1u << 3,
    RLGPU_CF_WHEEL3          = 1u << 4,
    RLGPU_CF_HAS_JUMPED      = 1u << 5,
    RLGPU_CF_HAS_DOUBLE_JUMPED = 1u << 6,
    RLGPU_CF_HAS_FLIPPED     = 1u << 7,
    RLGPU_CF_IS_FLIPPING     = 1u << 8,
    RLGPU_CF_IS_JUMPING      = 1u << 9,
    RLGPU_CF_IS_SUPERSONIC   = 1u << 10,
    RLGPU_CF_IS_AUTOFLIPPING = 1u << 11,
    RLGPU_CF_WORLD_CONTACT   = 1u << 12,
    RLGPU_CF_IS_DEMOED       = 1u << 13,
    RLGPU_CF_BALLHIT_VALID   = 1u << 14,
    RLGPU_CF_ABSENT          = 1u << 15  /* no car in this slot: the orange slots of an env created with one_team (spawnOpponents = false) */
};

typedef struct RlgpuCarState {
    float pos[3];
    float rot[9];            /* forward[3], right[3], up[3] */
    float vel[3];
    float ang_vel[3];
    uint32_t flags;          /* RLGPU_CF_* */
    float flip_rel_torque[3];
    float jump_time, flip_time;
    float air_time, air_time_since_jump;
    float boost;             /* 0..100 */
    float time_spent_boosting;
    float supersonic_time;
    float handbrake_val;
    float auto_flip_timer, auto_flip_torque_scale;
    float world_contact_normal[3];
    int32_t car_contact_other_id;   /* 0 = none; car id = slot+1 */
    float car_contact_cooldown;
    float demo_respawn_timer;
    /* BallHitInfo (BallHitInfo.h:10-25); ticks are absolute arena tick counts, -1 = never */
    float bh_rel_pos[3], bh_ball_pos[3], bh_extra_hit_vel[3];
    int64_t bh_tick_hit, bh_tick_extra;
    float last_controls[8];  /* throttle steer pitch yaw roll jump boost handbrake (bools as 0/1) */
    float controls[8];       /* the controls the next tick will use (Car::controls) */
    float vel_impulse_cache[3]; /* uu/s, applied at the end of the tick (Car.cpp:171-174) */
    float extra_pushback[4]; /* per-wheel, impulse units of the reference (kg*BT/s) */
    /* btWheelInfoRL values written by Car::_UpdateWheels in tick t and consumed by
     * btVehicleRL::updateVehicleFirst/calcFrictionImpulses in tick t+1 (Car.cpp:330-475,
     * btVehicleRL.cpp:64-92,313-387): they are carried state, and the reference does NOT reset
     * them in Car::SetState. */
    float wheel_steer_angle;      /* front wheels, radians */
    float wheel_engine_force;     /* all wheels */
    float wheel_brake;            /* all wheels */
    float wheel_lat_friction[4];
    float wheel_long_friction[4];
} RlgpuCarState;

typedef struct RlgpuBallState {
    float pos[3];
    float vel[3];
    float ang_vel[3];
    float vel_impulse_cache[3]; /* uu/s */
} RlgpuBallState;

typedef struct RlgpuPadState {
    float cooldown;
    uint8_t is_active;
    uint8_t _pad[3];
    int32_t prev_locked_car_id;  /* 0 = none */
} RlgpuPadState;

/* per-player gym-level carried state */
typedef struct RlgpuPlayerGymState {
    int32_t match_goals, match_saves, match_assists, match_shots, match_shot_passes,
            match_bumps, match_demos, boost_pickups;          /* PlayerData.h:17-25 */
    float event_last[RLGPU_NUM_EVENT_VALS];                   /* EventReward::lastRegisteredValues */
    float prev_action[8];                                     /* Match::prevActions row */
    int32_t prev_action_idx;                                  /* its index in the action table, -1 = zero action */
} RlgpuPlayerGymState;

typedef struct RlgpuGymState {
    int32_t score_line[2];           /* GameState.h:8-17 */
    int32_t last_touch_car_id;       /* GameState.h:24 */
    int64_t last_tick_count;         /* GameState.h:44 */
    int32_t no_touch_steps;          /* NoTouchCondition::stepsSinceTouch */
    /* GameEventTracker (GameEventTracker.h:78-86) */
    float   shot_cooldown;
    uint8_t ball_shot, ball_shot_goal_team, ball_scored_last, _pad0;
    int64_t last_ball_update_count;
    uint32_t snap_demoed_mask;       /* bit k: player k was demoed in the previous snapshot (Match::ParseActions, Match.cpp:47-49) */
    uint32_t episode_steps;          /* Gym::totalSteps since the last reset */
    uint32_t reset_count;            /* counter of the env's state-setter RNG stream */
    uint32_t _pad1;
    RlgpuPlayerGymState players[RLGPU_MAX_CARS];
} RlgpuGymState;

/* What the reference's arena holds besides the CarState / BallState fields above.  Appended in round 4 (older recordings of this struct are
 * simply shorter: readers pad them with ball_rot = identity, valid = 0).
 *   ball_rot   BallState::rotMat (Ball.h:17-44, Ball.cpp:27-30,41).  Under ArenaConfig::noBallRot (ArenaConfig.h:33, the default, which the
 *              gym's arenas use) the ball's orientation is never integrated (btRigidBody.cpp:102-106): it stays what the last SetState gave it
 *              and is reported back by every GetState.  rlgpu_env_upload_states stores it (all zeros = identity) in the env's resident
 *              words, every download hands it back.  The kernels and the host build of the stepper (oracle/arena_port.cpp) step in this
 *              basis, bit for bit like the reference (tests/golden/ballrot_golden.npz).
 *   valid      which of the two fields below mean something (RLGPU_HIDDEN_*).  Downloads set both bits; a struct with a bit clear (every
 *              recording, every state a user fills in) is Arena::SetState on an arena that keeps what it has: an env slot keeps its own
 *              broadphase history, a demolished car's body takes the reported basis (Car.cpp:22-36).
 *   bp_hist    btRSBroadphase's memory of its dynamic proxies ([0] ball, [1 + k] car slot k): cell of the last setAabb (bits 3..15) and arrival
 *              rank among the dynamic proxies (bits 0..2) -- which decide the order of the overlapping pairs, hence of the manifolds, hence
 *              of the solver's rows (csrc/arena_contact.h).  0 = never filed (a fresh arena).
 *   wreck_rot  forward / right / up of a DEMOLISHED car's rigid body: the body keeps turning while the car's state reports the rotation of the
 *              moment it was demolished (Car.cpp:69-80,135-138); all zeros for a car that is not demolished.
 *              With both, a state downloaded mid-episode and uploaded into another env slot (or another batch) continues exactly as the env it
 *              came from would have from that same struct (tests/golden/midtape_golden.npz: against the reference, whose side of this is read by
 *              oracle/ref_driver.cpp:ref_arena_get_hidden from btRSBroadphase's cell lists).
 *   ref_engine (round 6, parity tests) the state of the std::default_random_engine the reference's thread draws from (Math::GetRandEngine,
 *              RocketSim Math.cpp:59-64: libstdc++'s minstd_rand0, a value in [1, 2^31 - 2]) as it stands for this arena.  The reference seeds it from
 *              the wall clock, so no two of its runs draw alike and the product draws from counter-based streams of its own (Philox keyed by seed,
 *              env, tick / reset count: csrc/arena_car.h, arena_gym.h) -- ref_engine = 0, the default.  With a non-zero value the env draws
 *              EVERYTHING the reference draws for an arena -- the respawn slot of a demolished car (Car.cpp:43-56), ResetToRandomKickoff's shuffle
 *              (Arena.cpp:112-216), RandomState's values (RandomState.cpp:8-61) -- from that engine, with the reference's formulas, in the
 *              reference's order (car_order where it loops over the cars), and hands the advanced state back: an arena of the real reference whose
 *              engine was assigned the same state (oracle/ref_driver.cpp:ref_seed_engine) then makes the same draws, and tapes through respawns and
 *              setter outputs compare for EQUALITY (tests/golden/respawn_golden.npz, setter_golden.npz).  An upload without the valid bit leaves the
 *              slot's engine alone (Arena::SetState does not touch the thread's engine either). */
#define RLGPU_HIDDEN_BP_HIST   1u
#define RLGPU_HIDDEN_WRECK_ROT 2u
#define RLGPU_HIDDEN_REF_ENGINE 4u
#define RLGPU_HIDDEN_MUTATORS  8u   /* RlgpuArenaState::mutators is filled in */
typedef struct RlgpuArenaHidden {
    float ball_rot[9];               /* forward / right / up columns */
    uint32_t valid;
    uint16_t bp_hist[8];
    float wreck_rot[RLGPU_MAX_CARS][9];
    uint32_t ref_engine;             /* 0 = the env's own streams */
    uint32_t _pad;
} RlgpuArenaHidden;

/* MutatorConfig's run-time scalars (RocketSim MutatorConfig.h:18-75; round 6): what an env simulates with instead of RLConst's defaults.  The fields are the
 * reference's, in its units, except ball_damp_per_tick = powf(1 - ballDrag, 1 / 120) as the C library rounds it (btRigidBody::applyDamping computes that
 * every tick; the device has no bit-identical powf, so whoever fills the struct computes it on the host: rlgpu_default_mutators / the facade's
 * Arena::SetMutatorConfig).  NOT here, because they change a collision shape or a mass and stay compiled in: carMass, ballMass, ballRadius.  A state carries its env's block when hidden.valid has RLGPU_HIDDEN_MUTATORS (downloads
 * set it); an upload without the bit leaves the slot's mutators alone. */
#define RLGPU_MUT_UNLIMITED_FLIPS        1u
#define RLGPU_MUT_UNLIMITED_DOUBLE_JUMPS 2u
#define RLGPU_MUT_DEMO_ON_CONTACT        4u   /* DemoMode::ON_CONTACT */
#define RLGPU_MUT_DEMO_DISABLED          8u   /* DemoMode::DISABLED (neither bit: NORMAL) */
#define RLGPU_MUT_TEAM_DEMOS            16u   /* enableTeamDemos */
#define RLGPU_MUT_RAY_PROXY_LISTS       32u   /* NOT a MutatorConfig field.  Wheel rays are cast against every dynamic body the reference's broadphase lists for the ray's cell
                                               * (btRSBroadphase.cpp:326-337), not only those whose box the ray comes near: reproduces btSubsimplexConvexCast's occasional hit on a car
                                               * the ray misses by 20 - 30 uu (32 iterations run out).  Exact, 2.7 - 4.8 % slower; off by default (csrc/arena_world.h ray_ball_and_cars) */
typedef struct RlgpuMutators {
    float gravity_z;
    float boost_accel_ground, boost_accel_air, boost_used_per_second;
    float jump_accel, jump_immediate_force;
    float ball_max_speed;
    float ball_damp_per_tick;
    float respawn_delay, bump_cooldown_time, boost_pad_cooldown_big, boost_pad_cooldown_small;
    float car_spawn_boost_amount;
    float ball_hit_extra_force_scale, bump_force_scale;
    float goal_base_threshold_y;
    uint32_t flags;                  /* RLGPU_MUT_* */
    uint32_t _pad;
    float gravity_x, gravity_y;
    float car_world_friction, car_world_restitution;
    float ball_world_friction, ball_world_restitution;   /* the ball body's own values (Arena.cpp:44-45); against the static world Bullet takes min(f, 0.6) / max(r, 0.3) */
} RlgpuMutators;

typedef struct RlgpuArenaState {
    int32_t num_cars;                /* 2, 4 or 6 */
    uint32_t car_order;              /* the order in which the arena's per-car loops visit the cars: 4 bits per rank, slot + 1 (rank 0 in bits 0-3);
                                        0 = slot order.  The reference iterates `std::unordered_set<Car*> _cars` (Arena.h:35, Arena.cpp:716-812), i.e. in
                                        an order that follows from heap addresses; it matters only where one car's update reads another's (a wheel
                                        standing on a car, two cars on one boost pad).  States read from the reference carry its order. */
    int64_t tick_count;              /* Arena::tickCount */
    int64_t ball_update_counter;     /* BallState::updateCounter (reset by SetState) */
    RlgpuBallState ball;
    RlgpuCarState cars[RLGPU_MAX_CARS];
    RlgpuPadState pads[RLGPU_NUM_PADS];   /* RocketSim order: 6 big then 28 small (RLConst.h:210-253) */
    RlgpuGymState gym;
    RlgpuArenaHidden hidden;
    RlgpuMutators mutators;          /* meaningful when hidden.valid & RLGPU_HIDDEN_MUTATORS */
} RlgpuArenaState;

/* Step records (round 6): what a GameState is made of -- GameState::UpdateFromArena / PlayerData::UpdateFromCar, SIM/Utils/Gamestates/GameState.cpp:52-104,
 * PlayerData.cpp:4-34 -- and nothing else, for plugins whose work can wait until a collection launch has finished: a user RewardFunction, a step callback
 * (rlgpu_env_enable_step_records).  One record = RlgpuStepHead followed by num_cars x RlgpuStepCar, all 32-bit words; a 1v1 record is 336 bytes where a
 * full RlgpuArenaState snapshot is 3.7 KB.  Units as above (uu, rad/s; rot = forward / right / up). */
typedef struct RlgpuStepHead {
    int64_t tick_count;              /* Arena::tickCount where Gym::Step builds the step's GameState (after the first tick, Gym.cpp:81-93) */
    int32_t score_line[2];
    int32_t last_touch_car_id;
    uint32_t pads_active[2];         /* bit p of the 64: BoostPad p (RocketSim order) is active */
    float ball_pos[3], ball_vel[3], ball_ang_vel[3];
    uint32_t done;                   /* the step ended the episode (the terminal conditions' verdict); 0 in a reset record */
    int32_t num_cars;
} RlgpuStepHead;
typedef struct RlgpuStepCar {
    float pos[3], rot[9], vel[3], ang_vel[3];
    uint32_t flags;                  /* RLGPU_CF_* */
    float boost, air_time_since_jump, jump_time, flip_time, demo_respawn_timer;
    uint32_t touched;                /* bit 0: PlayerData::ballTouchedStep, bit 1: ballTouchedTick (PlayerData.cpp:20-30) */
    int32_t counters[8];             /* match goals, saves, assists, shots, shot passes, bumps, demos, boost pickups (PlayerData.h:17-25) */
} RlgpuStepCar;
#define RLGPU_STEP_HEAD_WORDS ((int)(sizeof(RlgpuStepHead) / 4))
#define RLGPU_STEP_CAR_WORDS ((int)(sizeof(RlgpuStepCar) / 4))
#define RLGPU_STEP_RECORD_WORDS(num_cars) (RLGPU_STEP_HEAD_WORDS + (num_cars) * RLGPU_STEP_CAR_WORDS)

#ifdef __cplusplus
}
#endif
#endif /* RLGPU_STATE_H */
