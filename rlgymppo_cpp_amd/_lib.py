"""ctypes binding of include/rlgpu.h (librlgpu.so, built in-tree by rlgymppo_cpp_amd/csrc/Makefile).

There is no CPU fallback: if the HIP library is missing or fails to load, importing the product path raises.
"""
import ctypes as C
import os

from .state import ArenaState

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RLGPU_LIB", os.path.join(_HERE, "librlgpu.so"))  # RLGPU_LIB: tuning experiments only

NUM_EVENT_VALS = 11

RW_EVENT, RW_VELOCITY, RW_SAVE_BOOST, RW_VEL_BALL_TO_GOAL, RW_VEL_PLAYER_TO_BALL, RW_FACE_BALL, RW_TOUCH_BALL = range(7)
TC_NO_TOUCH, TC_GOAL_SCORE = 0, 1
SS_RANDOM, SS_KICKOFF = 0, 1


class RewardTerm(C.Structure):
    _fields_ = [("kind", C.c_int32), ("weight", C.c_float), ("p0", C.c_float)]


class GymConfig(C.Structure):
    _fields_ = [
        ("tick_skip", C.c_int32),
        ("n_terms", C.c_int32), ("terms", RewardTerm * 8),
        ("event_weights", C.c_float * NUM_EVENT_VALS),
        ("zero_sum", C.c_int32), ("team_spirit", C.c_float), ("opp_scale", C.c_float),
        ("n_conds", C.c_int32), ("conds", C.c_int32 * 4), ("no_touch_max_steps", C.c_int32),
        ("setter_kind", C.c_int32), ("rand_ball_speed", C.c_int32), ("rand_car_speed", C.c_int32), ("cars_on_ground", C.c_int32),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32),
        ("pos_coef", C.c_float * 3), ("vel_coef", C.c_float), ("ang_vel_coef", C.c_float),
        ("n_actions", C.c_int32), ("obs_max_players", C.c_int32), ("one_team", C.c_int32), ("host_resets", C.c_int32),
    ]


class LearnerConfigC(C.Structure):
    _fields_ = [
        ("obs_size", C.c_int32), ("n_actions", C.c_int32),
        ("n_policy_layers", C.c_int32), ("policy_layers", C.c_int32 * 8),
        ("n_critic_layers", C.c_int32), ("critic_layers", C.c_int32 * 8),
        ("policy_lr", C.c_float), ("critic_lr", C.c_float), ("ent_coef", C.c_float), ("clip_range", C.c_float),
        ("temperature", C.c_float), ("use_bf16", C.c_int32),
        ("seed_lo", C.c_uint32), ("seed_hi", C.c_uint32), ("max_rows", C.c_int32),
    ]


_vp = C.c_void_p
_i = C.c_int
_f = C.c_float

# name -> (restype, argtypes); every symbol include/rlgpu.h declares
SIGNATURES = {
    "rlgpu_default_gym_config": (None, [C.POINTER(GymConfig)]),
    "rlgpu_env_create": (_i, [C.POINTER(_vp), _i, _i, _i, C.POINTER(GymConfig)]),
    "rlgpu_env_destroy": (None, [_vp]),
    "rlgpu_env_last_error": (C.c_char_p, [_vp]),
    "rlgpu_env_set_stream": (_i, [_vp, _vp]),
    "rlgpu_env_obs_size": (_i, [_vp]),
    "rlgpu_env_num_agents": (_i, [_vp]),
    "rlgpu_env_num_actions": (_i, [_vp]),
    "rlgpu_env_state_words": (_i, [_vp]),
    "rlgpu_env_check_redzones": (_i, [_vp]),
    "rlgpu_env_lost_contact_count": (_i, [_vp, C.POINTER(C.c_uint64), _i]),
    "rlgpu_env_big_layout_ticks": (_i, [_vp, C.POINTER(C.c_uint64), _i]),
    "rlgpu_env_set_collect_queue": (_i, [_vp, _i]),
    "rlgpu_learner_check_redzones": (_i, [_vp]),
    "rlgpu_env_debug_overrun": (_i, [_vp, _i, _i]),
    "rlgpu_state_word_counts": (_i, [_i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rlgpu_env_set_mesh": (_i, [_vp, _vp, _i, _vp, _i]),
    "rlgpu_env_set_procedural_mesh": (_i, [_vp]),
    "rlgpu_env_load_cmf_dir": (_i, [_vp, C.c_char_p]),
    "rlgpu_procedural_mesh": (_i, [_vp, _i, _vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "rlgpu_mesh_visit_order": (_i, [_vp, _i, _vp, _i, _vp]),
    "rlgpu_action_table": (_i, [_vp, _i]),
    "rlgpu_env_upload_states": (_i, [_vp, _vp, _vp, _i]),
    "rlgpu_env_download_states": (_i, [_vp, _vp, _vp, _i]),
    "rlgpu_env_reset": (_i, [_vp, _i, _vp]),
    "rlgpu_env_reset_envs": (_i, [_vp, _vp, _i, _i, _vp]),
    "rlgpu_env_enable_snapshots": (_i, [_vp, _i]),
    "rlgpu_env_download_snapshots": (_i, [_vp, _vp, _i, _i]),
    "rlgpu_env_enable_step_records": (_i, [_vp, _i]),
    "rlgpu_env_step_record_words": (_i, [_vp]),
    "rlgpu_env_download_step_records": (_i, [_vp, _i, _vp, _vp, _i, _vp]),
    "rlgpu_env_step_controls": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "rlgpu_pad_location": (_i, [_i, _vp, _vp]),
    "rlgpu_env_enable_step_stats": (_i, [_vp, _i]),
    "rlgpu_comm_rendezvous_path": (_i, [C.c_char_p, _i]),
    "rlgpu_env_overflow_counts": (_i, [_vp, _vp, _i]),
    "rlgpu_env_epa_counts": (_i, [_vp, _vp, _i]),
    "rlgpu_procedural_mesh_ex": (_i, [_i, C.c_float, _vp, _i, _vp, _i, C.POINTER(_i), C.POINTER(_i)]),
    "rlgpu_learner_inference_is_standalone": (_i, [_vp]),
    "rlgpu_env_enable_timing": (_i, [_vp, _i]),
    "rlgpu_learner_enable_timing": (_i, [_vp, _i]),
    "rlgpu_env_step_stats": (_i, [_vp, _vp, _i]),
    "rlgpu_env_step": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "rlgpu_env_physics_ticks": (_i, [_vp, _i]),
    "rlgpu_env_set_controls": (_i, [_vp, _vp]),
    "rlgpu_env_sync": (_i, [_vp]),
    "rlgpu_env_last_step_ms": (_i, [_vp, C.POINTER(_f)]),
    "rlgpu_learner_create": (_i, [C.POINTER(_vp), _i, C.POINTER(LearnerConfigC)]),
    "rlgpu_learner_destroy": (None, [_vp]),
    "rlgpu_learner_last_error": (C.c_char_p, [_vp]),
    "rlgpu_learner_set_stream": (_i, [_vp, _vp]),
    "rlgpu_learner_num_params": (C.c_int64, [_vp, _i]),
    "rlgpu_learner_get_params": (_i, [_vp, _i, _vp]),
    "rlgpu_learner_set_params": (_i, [_vp, _i, _vp]),
    "rlgpu_learner_get_grads": (_i, [_vp, _i, _vp]),
    "rlgpu_learner_grad_buffer": (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "rlgpu_learner_param_buffer": (_i, [_vp, C.POINTER(_vp), C.POINTER(C.c_int64)]),
    "rlgpu_learner_get_adam_state": (_i, [_vp, _vp, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rlgpu_learner_set_adam_state": (_i, [_vp, _vp, _vp, C.c_int64, C.c_int64]),
    "rlgpu_policy_act": (_i, [_vp, _vp, _i, _i, _vp, _vp, _vp]),
    "rlgpu_policy_probs": (_i, [_vp, _vp, _i, _vp]),
    "rlgpu_value_forward": (_i, [_vp, _vp, _i, _vp]),
    "rlgpu_gae": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _i, _vp, _vp, _vp]),
    "rlgpu_ppo_minibatch": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp]),
    "rlgpu_zero_grads": (_i, [_vp]),
    "rlgpu_clip_adam_step": (_i, [_vp, _f, _f]),
    "rlgpu_learner_set_lr": (_i, [_vp, _f, _f]),
    "rlgpu_learner_loss_scale": (_i, [_vp, _vp, _vp, _vp]),
    "rlgpu_learner_set_temperature": (_i, [_vp, _f]),
    "rlgpu_env_reseed": (_i, [_vp, C.c_uint32, C.c_uint32]),
    "rlgpu_default_mutators": (None, [_vp]),
    "rlgpu_ball_damp_per_tick": (_f, [_f]),
    "rlgpu_env_set_mutators": (_i, [_vp, _vp]),
    "rlgpu_learner_set_sampler": (_i, [_vp, C.c_uint32, C.c_uint32]),
    "rlgpu_learner_set_deterministic": (_i, [_vp, _i]),
    "rlgpu_learner_get_sampler": (_i, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rlgpu_comm_unique_id": (_i, [_vp]),
    "rlgpu_comm_init": (_i, [C.POINTER(_vp), _i, _i, _i, _vp]),
    "rlgpu_comm_init_env": (_i, [C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)]),
    "rlgpu_comm_destroy": (_i, [_vp]),
    "rlgpu_comm_rank": (_i, [_vp]),
    "rlgpu_comm_world": (_i, [_vp]),
    "rlgpu_comm_last_error": (C.c_char_p, [_vp]),
    "rlgpu_allreduce_grads": (_i, [_vp, _vp]),
    "rlgpu_comm_allreduce_f32": (_i, [_vp, _vp, C.c_int64, _vp]),
    "rlgpu_comm_broadcast": (_i, [_vp, _vp, C.c_int64, _i, _vp]),
    "rlgpu_learner_refresh_shadows": (_i, [_vp]),
    "rlgpu_comm_check": (_i, [_vp]),
    "rlgpu_comm_device": (_i, [_vp]),
    "rlgpu_learner_sync_from_rank0": (_i, [_vp, _vp]),
    "rlgpu_learner_param_checksum": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "rlgpu_learner_replicas_equal": (_i, [_vp, _vp, C.POINTER(_i)]),
    "rlgpu_learner_sync": (_i, [_vp]),
    "rlgpu_learner_last_gemm": (_i, [_vp, C.POINTER(_f), C.POINTER(C.c_double)]),
    "rlgpu_env_timing_total": (_i, [_vp, C.POINTER(_f), C.POINTER(_i), _i]),
    "rlgpu_learner_timing_total": (_i, [_vp, C.POINTER(_f), C.POINTER(C.c_double), C.POINTER(_i), _i]),
    "rlgpu_shuffler_create": (_i, [C.POINTER(_vp), C.c_uint32]),
    "rlgpu_shuffler_destroy": (None, [_vp]),
    "rlgpu_shuffler_next": (_i, [_vp, C.c_int64, _vp]),
    "rlgpu_shuffler_next_rows": (_i, [_vp, C.c_int, C.c_int, _vp]),
    "rlgpu_collect": (_i, [_vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "rlgpu_collect_free": (_i, [_vp, _vp, C.c_int, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "rlgpu_gae_ragged": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _f, _f, _f, _f, _i, _vp, _vp, _vp]),
    "rlgpu_shuffler_next_i32": (_i, [_vp, C.c_int64, _vp]),
    "rlgpu_shuffler_get_state": (_i, [_vp, C.c_char_p, _i]),
    "rlgpu_shuffler_set_state": (_i, [_vp, C.c_char_p]),
    "rlgpu_expbuf_create_ragged": (_i, [C.POINTER(_vp), C.c_int64, C.c_int, C.c_int, C.c_int64]),
    "rlgpu_expbuf_submit_ragged": (_i, [_vp, _vp, C.c_int64, C.POINTER(C.c_int)]),
    "rlgpu_expbuf_map_rows": (_i, [_vp, _vp, C.c_int64, _vp]),
    "rlgpu_expbuf_map_rows_dev": (_i, [_vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "rlgpu_traj_offsets": (_i, [_vp, _i, _i, _vp, _vp]),
    "rlgpu_expbuf_create": (_i, [C.POINTER(_vp), C.c_int64, C.c_int, C.c_int]),
    "rlgpu_expbuf_destroy": (None, [_vp]),
    "rlgpu_expbuf_num_slots": (_i, [_vp]),
    "rlgpu_expbuf_submit": (_i, [_vp, C.POINTER(C.c_int)]),
    "rlgpu_expbuf_size": (C.c_int64, [_vp]),
    "rlgpu_expbuf_shuffled_rows": (_i, [_vp, _vp, _vp]),
    "rlgpu_lt_write_model": (_i, [C.c_char_p, _vp, C.c_int, _vp]),
    "rlgpu_lt_read_model": (_i, [C.c_char_p, _vp, C.c_int, _vp]),
    "rlgpu_lt_write_adam": (_i, [C.c_char_p, _vp, C.c_int, C.c_float, _vp, _vp, C.c_int64]),
    "rlgpu_lt_read_adam": (_i, [C.c_char_p, _vp, C.c_int, _vp, _vp, C.POINTER(C.c_int64)]),
    "rlgpu_lt_last_error": (C.c_char_p, []),
}

_lib = None


def load():
    """Load librlgpu.so; raise (never fall back) when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"rlgymppo_cpp_amd: HIP library {LIB_PATH} is missing. Build it with "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RlgpuError(RuntimeError):
    pass


def default_gym_config() -> GymConfig:
    cfg = GymConfig()
    load().rlgpu_default_gym_config(C.byref(cfg))
    return cfg
