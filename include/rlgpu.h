/*
 * rlgpu.h — the C-ABI of the MI355X-native hot path (librlgpu.so, built from rlgymppo_cpp_amd/csrc/).
 *
 * Plain C: opaque handles, plain pointers and sizes, int status codes (0 = ok, <0 = error; the text is
 * available from rlgpu_last_error()).  No C++ exceptions cross this boundary and no torch types appear in
 * it: pointers documented as "device" are raw HBM addresses (from hipMalloc, or torch's tensor.data_ptr()).
 * All launches go to the context's stream (rlgpu_set_stream; default = the null stream).
 *
 * Each entry point names the reference interface it stands in for (file:line under /root/reference;
 * SIM = RLGymPPO_CPP/RLGymSim_CPP/src/RLGymSim_CPP, RS = RLGymPPO_CPP/RLGymSim_CPP/RocketSim/src,
 * PRIV/PUB = RLGymPPO_CPP/src/{private,public}/RLGymPPO_CPP).  INTEGRATION.md shows the binding a
 * maintainer of the reference would add on top of these.
 */
#ifndef RLGPU_H
#define RLGPU_H

#include <stddef.h>
#include <stdint.h>
#include "rlgpu_state.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rlgpu_env rlgpu_env;         /* N batched arenas + gym layer, resident on one GPU */
typedef struct rlgpu_learner rlgpu_learner; /* policy + critic MLPs, Adam state, PPO scratch, resident on one GPU */

enum { RLGPU_OK = 0, RLGPU_ERR_ARG = -1, RLGPU_ERR_HIP = -2, RLGPU_ERR_STATE = -3, RLGPU_ERR_NOMEM = -4 };

/* reward terms / terminal conditions / state setters with a device fast path (the built-ins of
 * SIM/Utils/RewardFunctions/CommonRewards.h:6-123, TerminalConditions/{NoTouch,GoalScore}Condition.h,
 * StateSetters/{RandomState.cpp:8-61,KickoffState.h:7-10}) */
enum { RLGPU_RW_EVENT = 0, RLGPU_RW_VELOCITY = 1, RLGPU_RW_SAVE_BOOST = 2, RLGPU_RW_VEL_BALL_TO_GOAL = 3,
       RLGPU_RW_VEL_PLAYER_TO_BALL = 4, RLGPU_RW_FACE_BALL = 5, RLGPU_RW_TOUCH_BALL = 6 };
enum { RLGPU_TC_NO_TOUCH = 0, RLGPU_TC_GOAL_SCORE = 1 };
enum { RLGPU_SS_RANDOM = 0, RLGPU_SS_KICKOFF = 1 };

typedef struct RlgpuRewardTerm { int32_t kind; float weight; float p0; } RlgpuRewardTerm;

/* What `EnvCreateFn` builds per env in the reference (examplemain.cpp:58-100): Match(reward, conditions, obs,
 * parser, setter, teamSize) + Gym(match, tickSkip).  Same field order as csrc/arena_gym.h GymConfig. */
typedef struct RlgpuGymConfig {
    int32_t tick_skip;
    int32_t n_terms; RlgpuRewardTerm terms[8];       /* CombinedReward (CombinedReward.h:6-53) */
    float event_weights[RLGPU_NUM_EVENT_VALS];       /* EventReward::WeightScales (CommonRewards.h:19-40) */
    int32_t zero_sum; float team_spirit, opp_scale;  /* ZeroSumReward (ZeroSumReward.cpp:3-29) */
    int32_t n_conds; int32_t conds[4]; int32_t no_touch_max_steps;
    int32_t setter_kind; int32_t rand_ball_speed, rand_car_speed, cars_on_ground;
    uint32_t seed_lo, seed_hi;
    float pos_coef[3], vel_coef, ang_vel_coef;       /* DefaultOBS ctor (DefaultOBS.h:11-15) */
    int32_t n_actions;                               /* DiscreteAction: 90 */
    int32_t obs_max_players;                         /* 0: DefaultOBS.  m > 0: DefaultOBSPadded(maxPlayers = m) (DefaultOBSPadded.cpp:3-66): m-1 teammate and
                                                      * m opponent blocks (zero blocks where there is no player), each list shuffled per observation;
                                                      * team_size <= m <= 4, row width 51 + 38 m */
    int32_t one_team;                                /* Match(..., spawnOpponents = false) (SIM/Envs/Match.h:40, Gym.cpp:45-49): team_size blue cars, no orange ones.
                                                      * Agents per env = team_size; the env's state keeps 2 * team_size slots, the odd ones flagged RLGPU_CF_ABSENT */
    int32_t host_resets;                             /* 1: the state setter runs on the host (a user StateSetter: GameInst.cpp:27-32 -> Gym::Reset -> Match::ResetState).  A step that ends
                                                      * an episode then leaves the env as the episode left it -- boost pads included, which the new episode's first GameState still
                                                      * shows (Match.cpp:55-69) -- and writes no observation rows for it; the host downloads the state, runs the setter, uploads and calls
                                                      * rlgpu_env_reset_envs(run_setter = 0).  0: the kernel resets the env itself with setter_kind */
} RlgpuGymConfig;

/* fills cfg with the examplemain.cpp:58-100 stack: 0.1 FaceBall + 0.5 VelPlayerToBall + 1.0 VelBallToGoal +
 * 50 Event{teamGoal 1, concede -1}; NoTouch(150) then GoalScore; DefaultOBS; RandomState(true,true,true); tickSkip 8 */
void rlgpu_default_gym_config(RlgpuGymConfig* cfg);

/* ---- environment batch : replaces ThreadAgentManager + N x (GameInst, Gym, Match, Arena)
 *      (PRIV/Threading/ThreadAgentManager.h:10-69, PUB/Threading/GameInst.cpp:3-38, SIM/Gym.cpp:40-102) ---- */
/* location (uu) of boost pad `pad` in RocketSim's order -- 6 big, then 28 small (RS/RLConst.h:215-253) -- for the host Arena facade */
int rlgpu_pad_location(int pad, float* pos_uu, int* is_big);
int rlgpu_env_create(rlgpu_env** out, int device, int n_envs, int team_size, const RlgpuGymConfig* cfg);
void rlgpu_env_destroy(rlgpu_env* e);
const char* rlgpu_env_last_error(const rlgpu_env* e);
/* MutatorConfig's run-time scalars for every env of the batch: replaces Arena::SetMutatorConfig on each of the reference's arenas (Gym's constructor,
 * SIM/Gym.cpp:40-44; RS/Sim/Arena/Arena.cpp:15-48).  A fresh batch runs RLConst's defaults (rlgpu_default_mutators); episode resets keep what was set; an
 * uploaded state that carries a block (RLGPU_HIDDEN_MUTATORS) replaces its env's.  Which of MutatorConfig's fields these are, and which stay compiled in:
 * rlgpu_state.h RlgpuMutators.  rlgpu_ball_damp_per_tick(ballDrag) = what the struct's ball_damp_per_tick has to hold for that drag. */
void rlgpu_default_mutators(RlgpuMutators* m);
float rlgpu_ball_damp_per_tick(float ball_drag);
int rlgpu_env_set_mutators(rlgpu_env* e, const RlgpuMutators* m);
/* New keys for the env batch's counter-based RNG streams (RandomState resets, respawn spots, padded-obs shuffles): a resumed run passes
 * (its shard seed, a fresh epoch number) so that it does not replay the resets of the run it continues. */
int rlgpu_env_reseed(rlgpu_env* e, uint32_t seed_lo, uint32_t seed_hi);
int rlgpu_env_set_stream(rlgpu_env* e, void* hip_stream);
int rlgpu_env_obs_size(const rlgpu_env* e);    /* OBSBuilder::BuildOBS(...).size() probe (PUB/Learner.cpp:99-109): 51+19*players, padded: 51+38*maxPlayers */
int rlgpu_env_num_agents(const rlgpu_env* e);  /* n_envs * players, players = 2 * team_size (team_size with one_team); agent row = env * players + slot (slot / 2 with one_team) */
int rlgpu_env_num_actions(const rlgpu_env* e);
int rlgpu_env_state_words(const rlgpu_env* e); /* resident 32-bit words per env */
/* Debug mode: with RLGPU_REDZONE=<bytes> in the environment at rlgpu_env_create, every persistent device buffer of the batch is followed by that
 * many guard bytes; this reads them back (after a device synchronise) and fails, naming the buffer, if a kernel wrote past one. */
int rlgpu_env_check_redzones(rlgpu_env* e);
int rlgpu_learner_check_redzones(rlgpu_learner* l);   /* the same for the learner's buffers (RLGPU_REDZONE at rlgpu_learner_create) */
int rlgpu_env_debug_overrun(rlgpu_env* e, int which, int past);   /* test hook: writes one guard byte of buffer `which`, so the check can be seen to fail */
/* the two counts that must agree (no GPU needed): words arena_visit visits for a team size, and word rows the kernels stage per env */
int rlgpu_state_word_counts(int team_size, int* visited, int* staged);

/* arena collision mesh: RocketSim::Init / InitFromMem (RS/RocketSim.cpp:70-212). verts in uu. */
int rlgpu_env_set_mesh(rlgpu_env* e, const float* verts_uu, int n_verts, const int32_t* tris, int n_tris);
int rlgpu_env_set_procedural_mesh(rlgpu_env* e);
int rlgpu_env_load_cmf_dir(rlgpu_env* e, const char* soccar_dir); /* collision_meshes/soccar/<name>.cmf */
/* host helpers (no GPU needed): the procedural soccar mesh and the DiscreteAction table */
int rlgpu_procedural_mesh(float* verts_uu, int cap_verts, int32_t* tris, int cap_tris, int* n_verts, int* n_tris);
/* The same arena at a chosen resolution (bench.py --mesh tessellated: a stand-in for the triangle counts of the game's own soccar meshes):
 * fillets of `fillet_segments` strips (4 = rlgpu_procedural_mesh), every edge longer than max_edge_uu split at its midpoint (0 = none).
 * With null buffers only the counts are returned. */
int rlgpu_procedural_mesh_ex(int fillet_segments, float max_edge_uu, float* verts_uu, int cap_verts, int32_t* tris, int cap_tris, int* n_verts, int* n_tris);
/* order_out[i] = input triangle that is collided i-th when everything overlaps: the visiting order of the reference's quantized BVH
   (btOptimizedBvh::build + walkStacklessQuantizedTreeCacheFriendly, btQuantizedBvh.cpp:116-277,655-674), which this library's mesh keeps */
int rlgpu_mesh_visit_order(const float* verts_uu, int n_verts, const int32_t* tris, int n_tris, int32_t* order_out);
int rlgpu_action_table(float* out_rows_x8, int cap_rows); /* DiscreteAction::DiscreteAction (SIM/Utils/ActionParsers/DiscreteAction.cpp:3-67) */

/* Car::SetState/GetState, Ball::SetState/GetState, BoostPad::SetState for whole envs (RS/Sim/Car/Car.cpp:9-36,
 * RS/Sim/Ball/Ball.cpp:27-49): the host StateSetter / user-plugin fallback path. env_ids NULL = envs 0..n-1.
 * An upload is a SetState on the env's arena, as in the reference: what the arena keeps outside its car / ball states -- the broadphase's
 * memory of where its proxies were filed and in which order they arrived (btRSBroadphase.cpp:185-203) -- stays as it is; an env that has
 * never ticked is a fresh arena.  A demolished car's rigid body takes the uploaded rotation (Car.cpp:22-36). */
int rlgpu_env_upload_states(rlgpu_env* e, const RlgpuArenaState* host_states, const int32_t* env_ids, int n);
int rlgpu_env_download_states(rlgpu_env* e, RlgpuArenaState* host_states, const int32_t* env_ids, int n);

/* Gym::Reset for every env (SIM/Gym.cpp:58-66). run_setter=0 keeps the uploaded physical state and only does the
 * episode bookkeeping. obs_dev: [num_agents x obs_size] fp32 device, may be NULL. */
int rlgpu_env_reset(rlgpu_env* e, int run_setter, float* obs_dev);
/* The same for the listed envs only (host list): what GameInst::Step does for one game whose episode ended (GameInst.cpp:27-32) when
 * the terminal conditions or the state setter run on the host.  Only the obs rows of those envs are written. */
int rlgpu_env_reset_envs(rlgpu_env* e, const int32_t* env_ids, int n, int run_setter, float* obs_dev);

/* Host-plugin fallback (user RewardFunction / OBSBuilder / TerminalCondition subclasses, step callbacks): when enabled, every
 * rlgpu_env_step also stores each env's arena as it stands where Gym::Step builds the step's GameState -- after the first tick and the
 * event tracker's update (SIM/Gym.cpp:81-93) -- in the exchange layout of rlgpu_state.h, and rlgpu_env_download_snapshots copies
 * envs [first_env, first_env + n) of that buffer to the host.  Off by default: 3.4 KB per env and step of extra HBM writes. */
int rlgpu_env_enable_snapshots(rlgpu_env* e, int on);
int rlgpu_env_download_snapshots(rlgpu_env* e, RlgpuArenaState* host_states, int first_env, int n);

/* Step records for plugins whose work can wait until a collection launch is over (round 6): a user RewardFunction -- in the reference it runs inside the
 * agent threads between steps (PRIV/Threading/ThreadAgent.cpp:100-160, SIM/Envs/Match.cpp:25-30), but nothing of it feeds the next action -- and step callbacks.
 * With a ring of t_cap steps enabled, rlgpu_collect / rlgpu_collect_free also store, per env and step, what a GameState is made of where Gym::Step builds it
 * (RlgpuStepHead + num_cars x RlgpuStepCar, rlgpu_state.h: 336 B for 1v1 against 3.7 KB of a full snapshot), and for every episode a launch ends the first
 * state of the next one (what the plugins' Reset hooks get).  rlgpu_env_download_step_records waits for the env's stream and copies steps [0, t_used) of
 * the ring -- host_ring [t_used][n_envs][record words] -- and the reset list: host_resets [n][2 + record words] = {env, step, record}, *n_resets = n
 * (RLGPU_ERR_ARG when n > reset_cap; n_envs * t_used is always enough); the list starts over with the next launch.  t_cap = 0 frees the ring. */
int rlgpu_env_enable_step_records(rlgpu_env* e, int t_cap);
int rlgpu_env_step_record_words(const rlgpu_env* e);
int rlgpu_env_download_step_records(rlgpu_env* e, int t_used, uint32_t* host_ring, uint32_t* host_resets, int reset_cap, int* n_resets);

/* The per-step player statistics the reference's example program gathers in its step callback (examplemain.cpp:23-36: speed, touch
 * ratio, airborne ratio), accumulated by the step kernels from every step's GameState so that they cost no host work:
 * out4 = {player-steps, sum of |car velocity| in uu/s, ball touches, airborne player-steps} since the last reset. */
int rlgpu_env_enable_step_stats(rlgpu_env* e, int on);
int rlgpu_env_step_stats(rlgpu_env* e, float* out4, int reset);

/* Gym::Step + GameInst auto-reset for every env (SIM/Gym.cpp:68-102, PUB/Threading/GameInst.cpp:7-38):
 * actions_dev [num_agents] int32 ; next_obs_dev [num_agents x obs_size] (post-reset obs when done);
 * reward_dev [num_agents] ; done_dev [num_agents] int32 (the env's done replicated to its players). */
int rlgpu_env_step(rlgpu_env* e, const int32_t* actions_dev, float* next_obs_dev, float* reward_dev, int32_t* done_dev);

/* The same step with the controls already parsed on the host -- a user ActionParser, or a standalone Gym (SIM/Gym.cpp:69-79 sets
 * car->controls from Match::ParseActions): controls_dev [num_agents x 8] fp32 device, one Action row per agent
 * {throttle, steer, pitch, yaw, roll, jump, boost, handbrake}.  The previous-action block of the device obs builder shows those rows. */
int rlgpu_env_step_controls(rlgpu_env* e, const float* controls_dev, float* next_obs_dev, float* reward_dev, int32_t* done_dev);

/* How often the narrowphase's fixed-size queues overflowed since the last reset (process-wide; every overflow sends that env through the
 * inline fallback for that tick -- same results, slower): out5 = {BVH frontier, ball candidate region, car candidate region, item queue, result pool}.
 * (Until round 4 the last one also counted the two cases in which a contact point is LOST; they have a counter of their own now, below.)
 * reset != 0: the counts as they stood are returned, then these five are cleared (the EPA counts below are not touched). */
int rlgpu_env_overflow_counts(rlgpu_env* e, uint64_t* out5, int reset);
/* Contact points LOST since the last reset (process-wide): always 0.  Until round 4 two cases lost points (a body touching a third mesh object with
 * points at once; a car-car point beyond the env's pair pool); such a tick is now detected before any contact callback has fired and the env's
 * world step is redone with a contact layout that has a manifold for every mesh object, a slot for every pair and a solver row for every slot
 * (csrc/arena_contact.h, arena_step.h:world_step_finish_big).  The counter stays as an invariant for tests and soak runs to assert.
 * reset != 0 clears it after reading. */
int rlgpu_env_lost_contact_count(rlgpu_env* e, uint64_t* out1, int reset);
/* env-ticks since the last reset (process-wide) that took that path: contacts beyond the LDS-resident layout, redone with the big one in global
 * memory -- same results as the reference, a few hundred microseconds each (two in 393 M env-ticks of learned 3v3).  reset != 0 clears it after reading. */
int rlgpu_env_big_layout_ticks(rlgpu_env* e, uint64_t* out1, int reset);

/* Penetration-depth queries since the last reset (process-wide): hitbox-mesh / hitbox-ball pairs whose cores overlap go through the
 * reference's second GJK + EPA (btGjkEpaPenetrationDepthSolver.cpp:24-79, btGjkEpa2.cpp; csrc/arena_epa.h).  out2 = {queries, queries that
 * did not fit the LDS arena and were repeated in the full-size one (Bullet's 128 vertices / 256 faces) in global memory}.
 * reset != 0: the counts as they stood are returned, then these two are cleared. */
int rlgpu_env_epa_counts(rlgpu_env* e, uint64_t* out2, int reset);

/* Arena::Step(ticks) on the resident states with the controls stored in them (RS/Sim/Arena/Arena.cpp:716-812) */
int rlgpu_env_physics_ticks(rlgpu_env* e, int ticks);
/* Car::controls of every car of every env from a host array [n_envs][2 * team_size][8] (throttle, steer, pitch, yaw, roll, jump, boost,
   handbrake: CarControls.h:6-24), nothing else touched: what `car->controls = c` does on the reference's Arena between Step calls.  The
   resident state is not rounded to uu and back (a download / upload pair would), so a tape of controls run this way through
   rlgpu_env_physics_ticks is the reference's free-running arena tick for tick. */
int rlgpu_env_set_controls(rlgpu_env* e, const float* controls_host);
int rlgpu_env_sync(rlgpu_env* e);
/* last rlgpu_env_step kernel duration in ms, measured with hipEvents on the context stream (bench.py roofline) */
/* Timing is opt-in (bench, profiling tools): without it the step / collect launches are not bracketed by events at all */
int rlgpu_env_enable_timing(rlgpu_env* e, int on);
int rlgpu_env_last_step_ms(rlgpu_env* e, float* ms);
/* sum of the rlgpu_env_step kernel durations (hipEvents on the context stream) and the launch count since the last
 * reset; synchronises the stream. */
int rlgpu_env_timing_total(rlgpu_env* e, float* total_ms, int* launches, int reset);

/* ---- learner : replaces PPOLearner + DiscretePolicy + ValueEstimator + ExperienceBuffer sampling + ComputeGAE
 *      (PRIV/PPO/PPOLearner.cpp:67-349, DiscretePolicy.cpp:7-75, ValueEstimator.cpp:6-27, PRIV/Util/TorchFuncs.cpp:5-52) */
typedef struct RlgpuLearnerConfig {
    int32_t obs_size, n_actions;
    int32_t n_policy_layers; int32_t policy_layers[8];   /* PPOLearnerConfig::policyLayerSizes (PUB/PPO/PPOLearnerConfig.h:7) */
    int32_t n_critic_layers; int32_t critic_layers[8];   /* criticLayerSizes (:8) */
    float policy_lr, critic_lr, ent_coef, clip_range;    /* :11-15 */
    float temperature;                                   /* DiscretePolicy temperature (DiscretePolicy.h:16) */
    int32_t use_bf16;                                    /* autocastLearn (PPOLearnerConfig.h:19): 1 = bf16 MFMA operands (the reference's autocast dtype,
                                                            FrameworkTorch.h:14), fp32 accumulate / master; 2 = fp16 operands in the PPO minibatch kernels with
                                                            a dynamic loss scale (gradscaler.hpp:26-34), flagship shape only; 0 = fp32 */
    uint32_t seed_lo, seed_hi;
    int32_t max_rows;                                    /* largest row count of any call (scratch sizing) */
} RlgpuLearnerConfig;

int rlgpu_learner_create(rlgpu_learner** out, int device, const RlgpuLearnerConfig* cfg);
void rlgpu_learner_destroy(rlgpu_learner* l);
const char* rlgpu_learner_last_error(const rlgpu_learner* l);
int rlgpu_learner_set_stream(rlgpu_learner* l, void* hip_stream);
/* parameter count and flat fp32 views (policy then critic; per layer: W[out][in] row-major, then b[out], the
 * state-dict order `0.weight,0.bias,2.weight,...` of the reference's Sequential) */
int64_t rlgpu_learner_num_params(const rlgpu_learner* l, int which /*0 policy, 1 critic, 2 both*/);
int rlgpu_learner_get_params(rlgpu_learner* l, int which, float* host_out);
int rlgpu_learner_set_params(rlgpu_learner* l, int which, const float* host_in);
int rlgpu_learner_get_grads(rlgpu_learner* l, int which, float* host_out);
/* device pointer to the contiguous fp32 gradient buffer (policy then critic) for the RCCL all-reduce */
int rlgpu_learner_grad_buffer(rlgpu_learner* l, float** dev_ptr, int64_t* n_floats);
int rlgpu_learner_param_buffer(rlgpu_learner* l, float** dev_ptr, int64_t* n_floats);
int rlgpu_learner_get_adam_state(rlgpu_learner* l, float* host_m, float* host_v, int64_t* step_policy, int64_t* step_critic);
int rlgpu_learner_set_adam_state(rlgpu_learner* l, const float* host_m, const float* host_v, int64_t step_policy, int64_t step_critic);

/* DiscretePolicy::GetAction (DiscretePolicy.cpp:51-62): probs = clamp(softmax(logits/T),1e-11,1); sample =
 * argmax(p / q), q ~ Exp(1) (== torch.multinomial) ; logp = log(p[a]).  noise_dev: optional recorded q
 * [rows x n_actions]; NULL = counter-based Philox (seed, call counter).  deterministic: argmax, logp = 0. */
int rlgpu_policy_act(rlgpu_learner* l, const float* obs_dev, int rows, int deterministic, const float* noise_dev,
                     int32_t* actions_dev, float* logp_dev);
/* raw policy outputs for tests: probs_dev [rows x n_actions] */
int rlgpu_policy_probs(rlgpu_learner* l, const float* obs_dev, int rows, float* probs_dev);
/* ValueEstimator::Forward (ValueEstimator.cpp:25-27) */
int rlgpu_value_forward(rlgpu_learner* l, const float* obs_dev, int rows, float* values_dev);

/* TorchFuncs::ComputeGAE (TorchFuncs.cpp:5-52) on time-major device arrays [T][n_agents]:
 *   rews, dones, truncs: [T x n]; values: [(T+1) x n] (row T = V(next state after the last step));
 *   out adv, targets, returns: [T x n].
 * next_value_mode 0 = reference-faithful: the batch is the agent-major concatenation of the n trajectories and
 *   `V[step+1]` of a trajectory's last step is the first value of the NEXT trajectory (quirk Q1), only the very
 *   last row uses values[T][n-1];  1 = per-agent bootstrap with values[T][j]. */
int rlgpu_gae(rlgpu_learner* l, const float* rews_dev, const float* dones_dev, const float* truncs_dev, const float* values_dev,
              int T, int n, float gamma, float lambda, float ret_std, float clip_range, int next_value_mode,
              float* adv_dev, float* targets_dev, float* returns_dev);

/* One minibatch of PPOLearner::Learn (PPOLearner.cpp:127-215): forward both nets on obs[idx], losses scaled by
 * batch_size_ratio (= MB / B), backward, gradients ACCUMULATED into the grad buffer.  idx_dev NULL = rows 0..n-1.
 * metrics_dev (optional, accumulated): [0] sum entropy, [1] sum KL, [2] sum clip fraction, [3] sum ratio,
 * [4] sum value loss (unscaled, per-minibatch means), [5] count of minibatches. */
int rlgpu_ppo_minibatch(rlgpu_learner* l, const float* obs_dev, const int32_t* actions_dev, const float* old_logp_dev,
                        const float* adv_dev, const float* targets_dev, const int32_t* idx_dev, int n,
                        float batch_size_ratio, float* metrics_dev);
/* The same for trajectories of their own lengths (free-running collection, rlgpu_collect_free): agent j's trajectory is rows 0 .. steps[j / players] - 1
 * of column j of the time-major arrays (row stride n), values has one more row per trajectory (V of the state after its last step).  The batch
 * ComputeGAE sees in the reference is the concatenation of the NON-EMPTY trajectories (ThreadAgentManager.cpp:47-60): with next_value_mode 0 the
 * row after a trajectory's last step is the first state of the next non-empty trajectory (quirk Q1), the batch's last row uses its own next state.
 * truncs_dev NULL = the collector's mark (1 - done on every trajectory's last row, ThreadAgentManager.cpp:55).  Rows beyond a trajectory are not written. */
int rlgpu_gae_ragged(rlgpu_learner* l, const float* rews_dev, const float* dones_dev, const float* truncs_dev, const float* values_dev, int n,
                     const int32_t* steps_dev /* [n / players] */, int players, float gamma, float lambda, float ret_std, float clip_range,
                     int next_value_mode, float* adv_dev, float* targets_dev, float* returns_dev);
int rlgpu_zero_grads(rlgpu_learner* l);
/* clip_grad_norm_(params, max_norm) per network then Adam (PPOLearner.cpp:273-288; torch defaults b1 .9 b2 .999 eps 1e-8).
 * grad_scale multiplies the gradients first (1/world_size after the all-reduce). */
int rlgpu_clip_adam_step(rlgpu_learner* l, float max_norm, float grad_scale);
int rlgpu_learner_set_lr(rlgpu_learner* l, float policy_lr, float critic_lr);
/* Deterministic-gradient mode (LearnerConfig::deterministicGradients).  The reference's gradients are sums in a fixed order (one stream,
 * PRIV/PPO/PPOLearner.cpp:205-215).  The fused minibatch kernels (bf16 / fp16 operands, layers up to 256 wide: the flagship shape) always sum
 * that way: every row slab's partial dW / db goes to its own buffer and the slabs are added in slab order.  The per-layer kernels (fp32 mode, wider
 * nets) split a minibatch's rows over blocks that add with fp32 atomics in the order they finish; on = 1 makes one block sum all rows of its
 * tile instead (slower, fixed order).  With lockstep collection a run is then a function of its seed, bit for bit, on one rank or several. */
int rlgpu_learner_set_deterministic(rlgpu_learner* l, int on);
/* fp16 mode (use_bf16 = 2): amp::GradScaler's state (PRIV/Util/gradscaler.hpp:26-34,162,291) -- the loss scale the NEXT minibatch's loss gradient is
 * multiplied by (rlgpu_ppo_minibatch leaves scale x gradient in the gradient buffer; rlgpu_clip_adam_step unscales BEFORE the clip, skips an
 * optimizer whose gradient norm is not finite and then halves the scale; 2000 clean steps double it), the clean steps counted so far, the steps
 * skipped so far.  Other modes: scale 1. */
int rlgpu_learner_loss_scale(rlgpu_learner* l, float* scale, int* growth_steps, int* skipped_steps);
/* bf16 mode: rebuild the bf16 weight copies the GEMMs read NOW, on the learner's stream (they are otherwise rebuilt lazily by the next
 * forward pass).  For LearnerConfig::collectionDuringLearn: the learning stream calls it after every optimizer step, so that inference on
 * the collection stream reads the live weights like the reference's agent threads do (ThreadAgent.cpp:72-103). */
int rlgpu_learner_refresh_shadows(rlgpu_learner* l);
/* collectionDuringLearn (LearnerConfig.h:46-50) puts rlgpu_policy_act / rlgpu_value_forward on another stream than the PPO epochs.  That is
 * only sound when inference does not share the learner's activation scratch with rlgpu_ppo_minibatch: 1 = it runs in the fused inference
 * kernel (bf16 mode, nets that fit its LDS), 0 = it does not and the host must keep collection and learning in sequence.  (The collector
 * reads the live weights either way, as the reference's does: PRIV/Threading/ThreadAgent.cpp:60-70 has no snapshot.) */
int rlgpu_learner_inference_is_standalone(const rlgpu_learner* l);
/* Action-sampler key: the Philox noise of the sampler is keyed on (seed, stream, call counter, row).  `stream` separates ranks that
 * share an init seed (identical parameters, independent exploration: rank r sets stream r); `call_ctr` is the number of act calls made
 * so far -- restored from a checkpoint so that a resumed run does not replay the first iterations' noise. */
int rlgpu_learner_set_sampler(rlgpu_learner* l, uint32_t stream, uint32_t call_ctr);
int rlgpu_learner_get_sampler(rlgpu_learner* l, uint32_t* stream, uint32_t* call_ctr);
int rlgpu_learner_set_temperature(rlgpu_learner* l, float temperature);   /* DiscretePolicy::temperature, set per call by InferUnit (InferUnit.cpp:68,95) */
int rlgpu_learner_sync(rlgpu_learner* l);
/* last ppo_minibatch GEMM time in ms + its flop count (bench.py roofline for the MFMA-bound kernels) */
int rlgpu_learner_last_gemm(rlgpu_learner* l, float* ms, double* flops);
/* the same accumulated over every rlgpu_ppo_minibatch call since the last reset; synchronises the stream */
int rlgpu_learner_enable_timing(rlgpu_learner* l, int on);   /* opt-in, like rlgpu_env_enable_timing */
int rlgpu_learner_timing_total(rlgpu_learner* l, float* total_ms, double* total_flops, int* calls, int reset);

/* ---- minibatch order : ExperienceBuffer::GetAllBatchesShuffled (PRIV/PPO/ExperienceBuffer.cpp:106-121): iota(n) then
 *      std::shuffle with a PERSISTENT std::default_random_engine(seed) (ExperienceBuffer.h:33, .cpp:7-9).  Host-side
 *      (the permutation is built on the CPU in the reference too); this library links the same libstdc++ algorithm. */
typedef struct rlgpu_shuffler rlgpu_shuffler;
int rlgpu_shuffler_create(rlgpu_shuffler** out, uint32_t seed);
void rlgpu_shuffler_destroy(rlgpu_shuffler* s);
int rlgpu_shuffler_next(rlgpu_shuffler* s, int64_t n, int64_t* perm_out);
int rlgpu_shuffler_next_i32(rlgpu_shuffler* s, int64_t n, int32_t* perm_out);   /* the same draw (same engine consumption), 32-bit entries */
/* the engine's state as text (operator<< / >> of std::default_random_engine): a host that draws a permutation AHEAD for a size it predicts
 * puts the engine back when the prediction fails */
int rlgpu_shuffler_get_state(const rlgpu_shuffler* s, char* buf, int cap);
int rlgpu_shuffler_set_state(rlgpu_shuffler* s, const char* buf);
/* the same draw for a [T][n_agents] time-major experience buffer whose LOGICAL order is agent-major (trajectory after trajectory, as
 * ExperienceBuffer holds it): rows_out[i] = (p % T) * n_agents + p / T for p = perm[i], B = T * n_agents entries, ready for idx_dev */
int rlgpu_shuffler_next_rows(rlgpu_shuffler* s, int T, int n_agents, int32_t* rows_out);

/* ---- fused collection.  The hot loop of ThreadAgent::_RunFunc (PRIV/Threading/ThreadAgent.cpp:58-163) for all envs and T steps in ONE
 *      launch: for t < T { policy->GetAction(obs[t]) -> actions[t], logp[t];  GameInst::Step(actions[t]) -> obs[t+1], reward[t], done[t] }.
 *      Every wavefront infers the actions of its own envs' agents and steps them without meeting the other wavefronts between steps
 *      (DESIGN.md 4.1).  Same results as T alternations of rlgpu_policy_act and rlgpu_env_step (same sampler counters; log-probs within one ulp).  All pointers
 *      are device memory, time-major: obs [T+1][n_agents][obs_size] with block 0 = the current observations, the rest [T][n_agents].
 *      Needs the learner in bf16 mode with a policy whose hidden width fits the in-kernel scratch (256-wide nets do); otherwise
 *      RLGPU_ERR_STATE and the caller alternates the two calls itself.  Launches on the env's stream. ---- */
int rlgpu_collect(rlgpu_env* e, rlgpu_learner* l, int T, float* obs_dev, int32_t* actions_dev, float* logp_dev, float* reward_dev, int32_t* done_dev,
                  int deterministic);
/* Lockstep collection of a batch with more wavefront-groups than the device keeps resident (BASELINE configs[3] / [4]) goes through a queue of
 * (step, group) tickets taken by as many wavefronts as fit the device (csrc/rlgpu_env.hip:k_env_collect_q): the launch no longer waits for the
 * slot that happened to get the slow groups.  Same results as the per-group launch, bit for bit.  mode: -1 = automatic (the default: the queue
 * when the groups do not all fit), 0 = never, 1 = always (tests run small batches through it). */
int rlgpu_env_set_collect_queue(rlgpu_env* e, int mode);

/* The same phase as the reference's agents run it: FREE.  ThreadAgentManager::CollectTimesteps (PRIV/Threading/ThreadAgentManager.cpp:16-82) lets every
 * agent thread step its games at its own pace and takes whatever each trajectory holds once the agents TOGETHER have `amount` timesteps; an agent
 * stops by itself after maxCollect / numThreads of them (ThreadAgent.cpp:57-59).  Here: every wavefront goes on stepping its envs until the launch's
 * shared counter reaches target_agent_steps (checked before every gym step), at most T_cap steps; steps_dev[env] = gym steps env made (its players'
 * trajectory length), so sum(steps) * players >= target (short of it only when every env ran into T_cap) and < target + num_agents.  A launch no longer ends with its slowest
 * wavefront.  Rows beyond an env's own count are not written.  An env's rows are those of the lockstep call's first steps_dev[env] steps, bit for bit
 * (same sampler counters, same stepper).  Needs every workgroup of the launch resident at once: RLGPU_ERR_STATE otherwise (and whenever rlgpu_collect
 * would return it) -- the caller then collects in lockstep.  obs [T_cap+1][n_agents][obs_size], the rest [T_cap][n_agents], steps_dev [n_envs] int32. */
int rlgpu_collect_free(rlgpu_env* e, rlgpu_learner* l, int T_cap, int64_t target_agent_steps, float* obs_dev, int32_t* actions_dev, float* logp_dev,
                       float* reward_dev, int32_t* done_dev, int32_t* steps_dev, int deterministic);

/* ---- ExperienceBuffer (PRIV/PPO/ExperienceBuffer.{h,cpp}): FIFO over rows with capacity max_rows.  Host bookkeeping only:
 *      every submitted iteration (T x n_agents rows, time-major) stays in one of `num_slots` device slots the caller owns; the
 *      library tracks which rows are still inside the FIFO (shift-left on overflow, :37-58; an addition larger than the buffer
 *      keeps its last max_rows rows, :32-35) and draws GetAllBatchesShuffled's permutation (:104-126) as device row numbers
 *      slot * T * n_agents + t * n_agents + agent. ---- */
typedef struct rlgpu_expbuf rlgpu_expbuf;
int rlgpu_expbuf_create(rlgpu_expbuf** out, int64_t max_rows, int T, int n_agents);
void rlgpu_expbuf_destroy(rlgpu_expbuf* b);
int rlgpu_expbuf_num_slots(const rlgpu_expbuf* b);
int rlgpu_expbuf_submit(rlgpu_expbuf* b, int* slot_out);   /* account for one more iteration; *slot_out = where the caller must store it */
int64_t rlgpu_expbuf_size(const rlgpu_expbuf* b);          /* curSize */
int rlgpu_expbuf_shuffled_rows(rlgpu_expbuf* b, rlgpu_shuffler* s, int32_t* rows_out /* [curSize] */);
/* Iterations whose trajectories have their own lengths (rlgpu_collect_free; what ExperienceBuffer::SubmitExperience gets in the reference: the
 * concatenation of ragged trajectories, Learner.cpp:694-702).  A device slot holds T_cap x n_agents rows, time-major; slots are provided for
 * max_rows / min_rows_per_iteration + 1 iterations (at most 16).  submit_ragged: agent_steps [n_agents] host = every trajectory's length; with
 * keep_last > 0 only the iteration's LAST keep_last logical rows join the FIFO (a multi-GPU run keeps every rank's FIFO the same size this way).
 * rlgpu_expbuf_shuffled_rows / _map_rows handle both kinds of iteration.  _map_rows: a permutation of [0, curSize) drawn elsewhere
 * (rlgpu_shuffler_next_i32) -> device rows; _map_rows_dev: the same on the device (all iterations in the FIFO must be ragged submissions),
 * traj_off_dev [num_slots][n_agents + 1] int32 = rlgpu_traj_offsets of the iteration in each slot. */
int rlgpu_expbuf_create_ragged(rlgpu_expbuf** out, int64_t max_rows, int T_cap, int n_agents, int64_t min_rows_per_iteration);
int rlgpu_expbuf_submit_ragged(rlgpu_expbuf* b, const int32_t* agent_steps, int64_t keep_last, int* slot_out);
int rlgpu_expbuf_map_rows(rlgpu_expbuf* b, const int32_t* perm, int64_t n, int32_t* rows_out);
int rlgpu_expbuf_map_rows_dev(rlgpu_expbuf* b, const int32_t* perm_dev, int64_t n, const int32_t* traj_off_dev, int32_t* rows_dev, void* hip_stream);
/* off_dev[a] = sum of steps_dev[a' / players] over agents a' < a, off_dev[n_agents] = all rows (one small launch on hip_stream) */
int rlgpu_traj_offsets(const int32_t* steps_dev, int n_agents, int players, int32_t* off_dev, void* hip_stream);

/* ---- checkpoint payloads (PRIV/PPO/PPOLearner.cpp:362-477): the TorchScript zip archives that torch::save(nn::Sequential)
 *      (PPO_POLICY.lt, PPO_CRITIC.lt) and optim::Adam::save (PPO_*_OPTIM.lt) write, read and written without libtorch, so
 *      checkpoints interchange with the reference.  Host-only.  dims[n_linear + 1] = {inputs, hidden..., outputs}; params and
 *      the Adam moments are flat in state-dict order (0.weight [out][in], 0.bias, 2.weight, ...).  Reading fails (RLGPU_ERR_ARG,
 *      text from rlgpu_lt_last_error) when the archive's shapes differ from dims ("Saved model has different size", :380-408). ---- */
int rlgpu_lt_write_model(const char* path, const int32_t* dims, int n_linear, const float* params);
int rlgpu_lt_read_model(const char* path, const int32_t* dims, int n_linear, float* params_out);
int rlgpu_lt_write_adam(const char* path, const int32_t* dims, int n_linear, float lr, const float* exp_avg, const float* exp_avg_sq, int64_t step);
int rlgpu_lt_read_adam(const char* path, const int32_t* dims, int n_linear, float* exp_avg, float* exp_avg_sq, int64_t* step);
const char* rlgpu_lt_last_error(void);

/* ---- multi-GPU: one process per GPU, envs sharded across ranks, ONE gradient all-reduce per optimizer step --------
 * Replaces nothing in the reference (it is single-GPU); stands where BASELINE config[2] asks for "RCCL grad all-reduce over xGMI".
 * rlgpu_comm is an RCCL communicator; the data path never leaves the C-ABI (no torch.distributed). */
typedef struct rlgpu_comm rlgpu_comm;
#define RLGPU_COMM_ID_BYTES 128
int rlgpu_comm_unique_id(void* id_out /* RLGPU_COMM_ID_BYTES */);                                  /* rank 0: ncclGetUniqueId */
int rlgpu_comm_init(rlgpu_comm** out, int device, int rank, int world, const void* id_bytes);      /* collective: ncclCommInitRank */
int rlgpu_comm_rendezvous_path(char* buf, int cap);   /* the file rlgpu_comm_init_env will use in this process's environment (diagnostics, tests) */
int rlgpu_comm_init_env(rlgpu_comm** out, int* rank_out, int* world_out);                          /* RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT + file rendezvous */
int rlgpu_comm_destroy(rlgpu_comm* c);
int rlgpu_comm_rank(const rlgpu_comm* c);
int rlgpu_comm_world(const rlgpu_comm* c);
int rlgpu_comm_device(const rlgpu_comm* c);   /* the device the communicator is bound to (LOCAL_RANK): the learner whose gradients it reduces must live there */
const char* rlgpu_comm_last_error(const rlgpu_comm* c);
/* sum over ranks of the learner's flat gradient buffer [policy | critic], in place, on the learner's stream; follow with
 * rlgpu_clip_adam_step(l, 1 / world ...) so the clip sees the global-batch gradient (PPOLearner.cpp:273-288 semantics) */
int rlgpu_allreduce_grads(rlgpu_learner* l, rlgpu_comm* c);
int rlgpu_comm_allreduce_f32(rlgpu_comm* c, float* dev_ptr, int64_t n, void* stream);
int rlgpu_comm_broadcast(rlgpu_comm* c, void* dev_ptr, int64_t bytes, int root, void* stream);     /* parameters / return statistics from rank 0 */
/* fail-fast: RLGPU_OK, or an error whose text (rlgpu_comm_last_error) names what went wrong with the ranks' exchange since the last look -- RCCL's
 * asynchronous error (ncclCommGetAsyncError; also polled before and after every collective above), or a peer's failure.  Hosts call it once per
 * iteration and END THE PROCESS on an error: the replicas can no longer be assumed equal; a restart is a fresh launch of every rank. */
int rlgpu_comm_check(rlgpu_comm* c);
/* Replica consistency.  sync_from_rank0: parameters, Adam moments and step counters of rank 0 replace every other rank's (after construction and after
 * loading a checkpoint: equal seeds and equal files make the replicas equal, this makes them equal whatever the files were).  param_checksum: a 64-bit
 * digest of the fp32 parameter bits; replicas_equal broadcasts rank 0's and compares: *equal_out = 0 on a rank whose parameters differ. */
int rlgpu_learner_sync_from_rank0(rlgpu_learner* l, rlgpu_comm* c);
int rlgpu_learner_param_checksum(rlgpu_learner* l, uint64_t* out);
int rlgpu_learner_replicas_equal(rlgpu_learner* l, rlgpu_comm* c, int* equal_out);
/* RLGPU_COMM_TRANSPORT=shm (tests): rlgpu_comm_init_env builds the communicator on a POSIX shared-memory segment instead of RCCL -- host-staged
 * all-reduce / broadcast with sums in rank order -- so that the ranks of a launch can share one device (RCCL refuses that) and a one-GPU box runs
 * the hosts' whole N > 1 path. */

#ifdef __cplusplus
}
#endif
#endif /* RLGPU_H */
