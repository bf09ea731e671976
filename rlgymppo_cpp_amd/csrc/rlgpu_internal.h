// rlgpu_internal.h -- what the two translation units of librlgpu.so share besides the public C-ABI (not installed, not exported).
#pragma once
#include "infer_device.h"
#include <cstdlib>
// Environment switches of the library come in two kinds.
//  * PATH SELECTORS pick one of several COMPLETE implementations of the same call (the per-layer kernels instead of the fused minibatch kernels, ...):
//    the tests run them against each other; every one does all of the work.  Read with std::getenv where they are used: RLGPU_NO_FUSED,
//    RLGPU_STRIPE, RLGPU_NO_FUSED_INFER, RLGPU_NO_VALUE_STRIPE (+ the diagnostics RLGPU_REDZONE and the rlgpu_comm_* launcher variables).
//  * EXPERIMENT switches change tile sizes, leave work out, or route debug stamps into live buffers (RLGPU_DW_DEBUG, RLGPU_FZ_DEBUG, RLGPU_DW_SLAB,
//    RLGPU_FUSED_CHUNK, RLGPU_FUSED_PROF, RLGPU_FUSED_STAMPS, RLGPU_FUSED_VALUE_ROWS, RLGPU_ONE_STREAM, RLGPU_EXPERIMENT_DYN_LDS, RLGPU_SCRATCH_FILL):
//    they exist only in a build with -DRLGPU_EXPERIMENTS (tools); the release library does not contain their names.
#ifdef RLGPU_EXPERIMENTS
#define RLGPU_EXPERIMENT_ENV(name) std::getenv(name)
#else
#define RLGPU_EXPERIMENT_ENV(name) (static_cast<const char*>(nullptr))
#endif
struct rlgpu_learner;
// Describe the learner's POLICY net for in-kernel inference (bf16 mode only) and reserve `n_calls` consecutive sampler counters:
// head->call_ctr is the first of them (a caller doing step t uses call_ctr + t), exactly what n_calls rlgpu_policy_act calls would
// have used.  The bf16 weight copies are refreshed on `stream` first.  Returns RLGPU_ERR_STATE when the net does not fit the
// single-wavefront kernel (fp32 mode, hidden layers wider than the LDS budget `max_buf_bytes` per activation buffer, > 128 actions).
int rlgpu_internal_policy_net(rlgpu_learner* l, rlinfer::InferNet* net, rlinfer::HeadArgs* head, int deterministic, int n_calls, int max_buf_bytes, void* stream);
