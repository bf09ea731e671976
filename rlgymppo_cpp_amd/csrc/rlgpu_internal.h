// rlgpu_internal.h -- what the two translation units of librlgpu.so share besides the public C-ABI (not installed, not exported).
#pragma once
#include "infer_device.h"
struct rlgpu_learner;
// Describe the learner's POLICY net for in-kernel inference (bf16 mode only) and reserve `n_calls` consecutive sampler counters:
// head->call_ctr is the first of them (a caller doing step t uses call_ctr + t), exactly what n_calls rlgpu_policy_act calls would
// have used.  The bf16 weight copies are refreshed on `stream` first.  Returns RLGPU_ERR_STATE when the net does not fit the
// single-wavefront kernel (fp32 mode, hidden layers wider than the LDS budget `max_buf_bytes` per activation buffer, > 128 actions).
int rlgpu_internal_policy_net(rlgpu_learner* l, rlinfer::InferNet* net, rlinfer::HeadArgs* head, int deterministic, int n_calls, int max_buf_bytes, void* stream);
