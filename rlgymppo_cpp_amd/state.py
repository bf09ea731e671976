"""ctypes mirror of include/rlgpu_state.h (the AoS host exchange layout of one arena).

Field meaning and reference citations live in the C header; this file only restates the layout so
Python callers (tests, bench, the host-side Learner mirror) can build and read states.
"""
import ctypes as C
import numpy as np

MAX_CARS = 6
NUM_PADS = 34
NUM_EVENT_VALS = 11

CF_ON_GROUND = 1 << 0
CF_WHEEL0 = 1 << 1
CF_HAS_JUMPED = 1 << 5
CF_HAS_DOUBLE_JUMPED = 1 << 6
CF_HAS_FLIPPED = 1 << 7
CF_IS_FLIPPING = 1 << 8
CF_IS_JUMPING = 1 << 9
CF_IS_SUPERSONIC = 1 << 10
CF_IS_AUTOFLIPPING = 1 << 11
CF_WORLD_CONTACT = 1 << 12
CF_IS_DEMOED = 1 << 13
CF_BALLHIT_VALID = 1 << 14
CF_ABSENT = 1 << 15   # no car in this slot (the orange slots of a one-team env)

f32 = C.c_float


class CarState(C.Structure):
    _fields_ = [
        ("pos", f32 * 3), ("rot", f32 * 9), ("vel", f32 * 3), ("ang_vel", f32 * 3),
        ("flags", C.c_uint32),
        ("flip_rel_torque", f32 * 3),
        ("jump_time", f32), ("flip_time", f32), ("air_time", f32), ("air_time_since_jump", f32),
        ("boost", f32), ("time_spent_boosting", f32), ("supersonic_time", f32), ("handbrake_val", f32),
        ("auto_flip_timer", f32), ("auto_flip_torque_scale", f32),
        ("world_contact_normal", f32 * 3),
        ("car_contact_other_id", C.c_int32), ("car_contact_cooldown", f32), ("demo_respawn_timer", f32),
        ("bh_rel_pos", f32 * 3), ("bh_ball_pos", f32 * 3), ("bh_extra_hit_vel", f32 * 3),
        ("bh_tick_hit", C.c_int64), ("bh_tick_extra", C.c_int64),
        ("last_controls", f32 * 8), ("controls", f32 * 8),
        ("vel_impulse_cache", f32 * 3), ("extra_pushback", f32 * 4),
        ("wheel_steer_angle", f32), ("wheel_engine_force", f32), ("wheel_brake", f32),
        ("wheel_lat_friction", f32 * 4), ("wheel_long_friction", f32 * 4),
    ]


class BallState(C.Structure):
    _fields_ = [("pos", f32 * 3), ("vel", f32 * 3), ("ang_vel", f32 * 3), ("vel_impulse_cache", f32 * 3)]


class PadState(C.Structure):
    _fields_ = [("cooldown", f32), ("is_active", C.c_uint8), ("_pad", C.c_uint8 * 3), ("prev_locked_car_id", C.c_int32)]


class PlayerGymState(C.Structure):
    _fields_ = [
        ("match_goals", C.c_int32), ("match_saves", C.c_int32), ("match_assists", C.c_int32),
        ("match_shots", C.c_int32), ("match_shot_passes", C.c_int32), ("match_bumps", C.c_int32),
        ("match_demos", C.c_int32), ("boost_pickups", C.c_int32),
        ("event_last", f32 * NUM_EVENT_VALS), ("prev_action", f32 * 8), ("prev_action_idx", C.c_int32),
    ]


class GymState(C.Structure):
    _fields_ = [
        ("score_line", C.c_int32 * 2), ("last_touch_car_id", C.c_int32),
        ("last_tick_count", C.c_int64), ("no_touch_steps", C.c_int32),
        ("shot_cooldown", f32),
        ("ball_shot", C.c_uint8), ("ball_shot_goal_team", C.c_uint8), ("ball_scored_last", C.c_uint8), ("_pad0", C.c_uint8),
        ("last_ball_update_count", C.c_int64),
        ("snap_demoed_mask", C.c_uint32), ("episode_steps", C.c_uint32), ("reset_count", C.c_uint32), ("_pad1", C.c_uint32),
        ("players", PlayerGymState * MAX_CARS),
    ]


class ArenaHidden(C.Structure):
    """RlgpuArenaHidden (include/rlgpu_state.h): the ball's basis (BallState::rotMat); the broadphase's memory of its dynamic proxies (bp_hist) and a
    demolished car's own rigid-body basis (wreck_rot), valid bit 0 / bit 1 saying which of the two mean something (downloads set both); ref_engine
    (valid bit 2, parity tests): the state of the reference thread's std::default_random_engine this env draws from instead of its own streams, 0 = off."""
    _fields_ = [("ball_rot", f32 * 9), ("valid", C.c_uint32), ("bp_hist", C.c_uint16 * 8), ("wreck_rot", (f32 * 9) * MAX_CARS), ("ref_engine", C.c_uint32), ("_pad", C.c_uint32)]


HIDDEN_BP_HIST, HIDDEN_WRECK_ROT, HIDDEN_REF_ENGINE, HIDDEN_MUTATORS = 1, 2, 4, 8


class Mutators(C.Structure):
    """RlgpuMutators (include/rlgpu_state.h): MutatorConfig's run-time scalars; meaningful in a state when hidden.valid has HIDDEN_MUTATORS.
    ball_damp_per_tick = powf(1 - ballDrag, 1 / 120) by the C library (rlgpu_ball_damp_per_tick)."""
    _fields_ = [("gravity_z", f32), ("boost_accel_ground", f32), ("boost_accel_air", f32), ("boost_used_per_second", f32), ("jump_accel", f32), ("jump_immediate_force", f32),
                ("ball_max_speed", f32), ("ball_damp_per_tick", f32), ("respawn_delay", f32), ("bump_cooldown_time", f32), ("boost_pad_cooldown_big", f32),
                ("boost_pad_cooldown_small", f32), ("car_spawn_boost_amount", f32), ("ball_hit_extra_force_scale", f32), ("bump_force_scale", f32),
                ("goal_base_threshold_y", f32), ("flags", C.c_uint32), ("_pad", C.c_uint32), ("gravity_x", f32), ("gravity_y", f32),
                ("car_world_friction", f32), ("car_world_restitution", f32), ("ball_world_friction", f32), ("ball_world_restitution", f32)]


MUT_UNLIMITED_FLIPS, MUT_UNLIMITED_DOUBLE_JUMPS, MUT_DEMO_ON_CONTACT, MUT_DEMO_DISABLED, MUT_TEAM_DEMOS, MUT_RAY_PROXY_LISTS = 1, 2, 4, 8, 16, 32


IDENTITY9 = (1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0)


class ArenaState(C.Structure):
    _fields_ = [
        ("num_cars", C.c_int32), ("car_order", C.c_uint32),
        ("tick_count", C.c_int64), ("ball_update_counter", C.c_int64),
        ("ball", BallState), ("cars", CarState * MAX_CARS), ("pads", PadState * NUM_PADS), ("gym", GymState),
        ("hidden", ArenaHidden), ("mutators", Mutators),
    ]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.hidden.ball_rot[:] = IDENTITY9      # a BallState's default rotMat

    @classmethod
    def from_buffer_copy(cls, buf, offset=0):
        """Also takes a recording made before the `hidden` block was appended (the committed fixtures): the ball's basis is the identity there
        and nothing hidden travels (valid = 0), which is what those recordings meant."""
        b = bytes(buf)[offset:]
        short = C.sizeof(cls) - C.sizeof(ArenaHidden) - C.sizeof(Mutators)
        if len(b) == short:
            out = type(C.Structure).from_buffer_copy(cls, b + bytes(C.sizeof(cls) - short))
            out.hidden.ball_rot[:] = IDENTITY9
            return out
        if short < len(b) < C.sizeof(cls):     # a recording of rounds 4 - 5 (the block ended with wreck_rot: no ref_engine, 0 = the env's own streams) or of
            b = b + bytes(C.sizeof(cls) - len(b))   # early round 6 (no mutators block: valid bit clear = RLConst's defaults)
        return type(C.Structure).from_buffer_copy(cls, b)


def yaw_rot(yaw: float):
    """forward/right/up columns of Angle(yaw,0,0).ToRotMat() (MathTypes.cpp:84-89)."""
    c, s = float(np.cos(yaw)), float(np.sin(yaw))
    return [c, s, 0.0, -s, c, 0.0, 0.0, 0.0, 1.0]


def euler_rot(yaw: float, pitch: float, roll: float):
    """RocketSim Angle(yaw,pitch,roll).ToRotMat() = btMatrix3x3::setEulerYPR(yaw,-pitch,-roll), returned as
    forward/right/up columns (MathTypes.cpp:84-89; btMatrix3x3.h setEulerZYX)."""
    ez, ey, ex = yaw, -pitch, -roll  # setEulerYPR(yaw,pitch,roll) -> setEulerZYX(roll, pitch, yaw)
    ci, cj, ch = np.cos(ex), np.cos(ey), np.cos(ez)
    si, sj, sh = np.sin(ex), np.sin(ey), np.sin(ez)
    cc, cs, sc, ss = ci * ch, ci * sh, si * ch, si * sh
    m = np.array([[cj * ch, sj * sc - cs, sj * cc + ss],
                  [cj * sh, sj * ss + cc, sj * cs - sc],
                  [-sj, cj * si, cj * ci]], dtype=np.float32)
    return [float(x) for x in (m[0, 0], m[1, 0], m[2, 0], m[0, 1], m[1, 1], m[2, 1], m[0, 2], m[1, 2], m[2, 2])]


def default_car(slot: int, pos=(0.0, 0.0, 17.0), yaw=0.0, boost=33.333333) -> CarState:
    cs = CarState()
    cs.pos[:] = pos
    cs.rot[:] = yaw_rot(yaw)
    cs.flags = CF_ON_GROUND
    cs.boost = boost
    cs.bh_tick_hit = -1
    cs.bh_tick_extra = -1
    return cs


def default_arena(num_cars=2) -> ArenaState:
    s = ArenaState()
    s.num_cars = num_cars
    s.ball.pos[:] = (0.0, 0.0, 93.15)
    for i in range(num_cars):
        blue = (i % 2 == 0)
        s.cars[i] = default_car(i, pos=(0.0, -2000.0 if blue else 2000.0, 17.0), yaw=np.pi / 2 if blue else -np.pi / 2)
    for p in s.pads:
        p.is_active = 1
    return s
