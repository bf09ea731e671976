"""PPOCore — host handle of the on-device learner (rlgpu_learner_* / rlgpu_policy_* / rlgpu_gae / rlgpu_ppo_* in
include/rlgpu.h).  Stands where the reference has PPOLearner + DiscretePolicy + ValueEstimator + TorchFuncs::ComputeGAE
(RLGymPPO_CPP/src/private/RLGymPPO_CPP/PPO/PPOLearner.cpp:67-349, DiscretePolicy.cpp:7-75, ValueEstimator.cpp:6-27,
Util/TorchFuncs.cpp:5-52).  torch only owns buffers / streams and provides torch.distributed for the gradient all-reduce.
"""
import ctypes as C
import numpy as np
import torch

from . import _lib


def _chk(rc, handle, errfn):
    if rc != 0:
        raise _lib.RlgpuError(f"rlgpu error {rc}: {errfn(handle).decode()}")


class PPOCore:
    def __init__(self, obs_size, n_actions, policy_layers=(256, 256, 256), critic_layers=(256, 256, 256), policy_lr=3e-4, critic_lr=3e-4,
                 ent_coef=0.005, clip_range=0.2, temperature=1.0, use_bf16=False, seed=123, max_rows=65536, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("PPOCore needs a GPU (all learner math is HIP kernels; there is no CPU path)")
        self.lib = _lib.load()
        c = _lib.LearnerConfigC()
        c.obs_size = obs_size; c.n_actions = n_actions
        c.n_policy_layers = len(policy_layers); c.n_critic_layers = len(critic_layers)
        for i, w in enumerate(policy_layers): c.policy_layers[i] = w
        for i, w in enumerate(critic_layers): c.critic_layers[i] = w
        c.policy_lr = policy_lr; c.critic_lr = critic_lr; c.ent_coef = ent_coef; c.clip_range = clip_range
        c.temperature = temperature; c.use_bf16 = 2 if use_bf16 == "fp16" else (1 if use_bf16 else 0)   # "fp16": fp16 operands + dynamic loss scale (include/rlgpu.h)
        c.seed_lo = seed & 0xffffffff; c.seed_hi = (seed >> 32) & 0xffffffff; c.max_rows = max_rows
        self.cfg = c
        self.device = device
        self.obs_size, self.n_actions, self.max_rows = obs_size, n_actions, max_rows
        self.policy_layers, self.critic_layers = tuple(policy_layers), tuple(critic_layers)
        self.h = C.c_void_p()
        _chk(self.lib.rlgpu_learner_create(C.byref(self.h), device, C.byref(c)), self.h, self.lib.rlgpu_learner_last_error)
        self.lib.rlgpu_learner_enable_timing(self.h, 1)   # test / tools host: the GEMM-section timers are read there
        self._err = self.lib.rlgpu_learner_last_error

    def close(self):
        if self.h:
            self.lib.rlgpu_learner_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _dev(self):
        return torch.device("cuda", self.device)

    def set_stream(self, stream):
        _chk(self.lib.rlgpu_learner_set_stream(self.h, C.c_void_p(stream.cuda_stream if stream is not None else 0)), self.h, self._err)

    def set_sampler(self, stream: int, call_ctr: int = 0):
        """Action-sampler key (rlgpu_learner_set_sampler): `stream` separates ranks that share the init seed, `call_ctr` resumes the noise sequence."""
        _chk(self.lib.rlgpu_learner_set_sampler(self.h, C.c_uint32(stream & 0xffffffff), C.c_uint32(call_ctr & 0xffffffff)), self.h, self._err)

    def get_sampler(self):
        s, c = C.c_uint32(), C.c_uint32()
        _chk(self.lib.rlgpu_learner_get_sampler(self.h, C.byref(s), C.byref(c)), self.h, self._err)
        return int(s.value), int(c.value)

    def refresh_shadows(self):
        _chk(self.lib.rlgpu_learner_refresh_shadows(self.h), self.h, self._err)

    # ---- parameters -------------------------------------------------------------------------------------------
    def num_params(self, which=2):
        return int(self.lib.rlgpu_learner_num_params(self.h, which))

    def get_params(self, which=2):
        a = np.empty(self.num_params(which), np.float32)
        _chk(self.lib.rlgpu_learner_get_params(self.h, which, a.ctypes.data), self.h, self._err)
        return a

    def set_params(self, arr, which=2):
        a = np.ascontiguousarray(arr, np.float32)
        assert a.size == self.num_params(which)
        _chk(self.lib.rlgpu_learner_set_params(self.h, which, a.ctypes.data), self.h, self._err)

    def get_grads(self, which=2):
        a = np.empty(self.num_params(which), np.float32)
        _chk(self.lib.rlgpu_learner_get_grads(self.h, which, a.ctypes.data), self.h, self._err)
        return a

    def get_adam_state(self):
        n = self.num_params(2)
        m = np.empty(n, np.float32); v = np.empty(n, np.float32)
        sp, sc = C.c_int64(), C.c_int64()
        _chk(self.lib.rlgpu_learner_get_adam_state(self.h, m.ctypes.data, v.ctypes.data, C.byref(sp), C.byref(sc)), self.h, self._err)
        return m, v, sp.value, sc.value

    def set_adam_state(self, m, v, step_p, step_c):
        m = np.ascontiguousarray(m, np.float32); v = np.ascontiguousarray(v, np.float32)
        _chk(self.lib.rlgpu_learner_set_adam_state(self.h, m.ctypes.data, v.ctypes.data, step_p, step_c), self.h, self._err)

    def grad_tensor(self) -> torch.Tensor:
        """A torch view (no copy) of the contiguous device gradient buffer, for torch.distributed.all_reduce."""
        p, n = C.c_void_p(), C.c_int64()
        _chk(self.lib.rlgpu_learner_grad_buffer(self.h, C.byref(p), C.byref(n)), self.h, self._err)
        return _wrap_device_f32(p.value, n.value, self.device)

    def param_tensor(self) -> torch.Tensor:
        p, n = C.c_void_p(), C.c_int64()
        _chk(self.lib.rlgpu_learner_param_buffer(self.h, C.byref(p), C.byref(n)), self.h, self._err)
        return _wrap_device_f32(p.value, n.value, self.device)

    def layer_shapes(self, which):
        """[(W shape, b shape)] in state-dict order 0.weight,0.bias,2.weight,...."""
        dims = [self.obs_size] + list(self.policy_layers if which == 0 else self.critic_layers) + [self.n_actions if which == 0 else 1]
        return [((dims[i + 1], dims[i]), (dims[i + 1],)) for i in range(len(dims) - 1)]

    # ---- inference --------------------------------------------------------------------------------------------
    def act(self, obs: torch.Tensor, actions: torch.Tensor, logp: torch.Tensor, deterministic=False, noise: "torch.Tensor | None" = None):
        rows = obs.shape[0]
        _chk(self.lib.rlgpu_policy_act(self.h, obs.data_ptr(), rows, 1 if deterministic else 0, noise.data_ptr() if noise is not None else None,
                                       actions.data_ptr(), logp.data_ptr()), self.h, self._err)

    def probs(self, obs: torch.Tensor) -> torch.Tensor:
        out = torch.empty((obs.shape[0], self.n_actions), dtype=torch.float32, device=obs.device)
        _chk(self.lib.rlgpu_policy_probs(self.h, obs.data_ptr(), obs.shape[0], out.data_ptr()), self.h, self._err)
        return out

    def value(self, obs: torch.Tensor, out: "torch.Tensor | None" = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((obs.shape[0],), dtype=torch.float32, device=obs.device)
        _chk(self.lib.rlgpu_value_forward(self.h, obs.data_ptr(), obs.shape[0], out.data_ptr()), self.h, self._err)
        return out

    # ---- GAE ------------------------------------------------------------------------------------------------------
    def gae(self, rews, dones, truncs, values, gamma, lam, ret_std, clip_range, next_value_mode=0):
        T, n = rews.shape
        assert values.shape == (T + 1, n)
        adv = torch.empty_like(rews); tgt = torch.empty_like(rews); ret = torch.empty_like(rews)
        _chk(self.lib.rlgpu_gae(self.h, rews.data_ptr(), dones.data_ptr(), truncs.data_ptr(), values.data_ptr(), T, n, gamma, lam, ret_std, clip_range,
                                next_value_mode, adv.data_ptr(), tgt.data_ptr(), ret.data_ptr()), self.h, self._err)
        return adv, tgt, ret

    # ---- learning -------------------------------------------------------------------------------------------------
    def zero_grads(self):
        _chk(self.lib.rlgpu_zero_grads(self.h), self.h, self._err)

    def minibatch(self, obs, actions, old_logp, adv, targets, idx, n, batch_size_ratio, metrics=None):
        _chk(self.lib.rlgpu_ppo_minibatch(self.h, obs.data_ptr(), actions.data_ptr(), old_logp.data_ptr(), adv.data_ptr(), targets.data_ptr(),
                                          idx.data_ptr() if idx is not None else None, n, batch_size_ratio,
                                          metrics.data_ptr() if metrics is not None else None), self.h, self._err)

    def clip_adam_step(self, max_norm=0.5, grad_scale=1.0):
        _chk(self.lib.rlgpu_clip_adam_step(self.h, max_norm, grad_scale), self.h, self._err)

    def check_redzones(self) -> None:
        """Debug mode (RLGPU_REDZONE=<bytes> in the environment when the learner was created): raises, naming the buffer, if a kernel wrote past the end of
        one of the learner's device buffers."""
        _chk(self.lib.rlgpu_learner_check_redzones(self.h), self.h, self._err)

    def loss_scale(self):
        """(scale, clean steps counted, steps skipped) of the fp16 mode's dynamic loss scale; scale 1 in the other modes."""
        import ctypes as C
        sc, g, k = C.c_float(), C.c_int(), C.c_int()
        _chk(self.lib.rlgpu_learner_loss_scale(self.h, C.byref(sc), C.byref(g), C.byref(k)), self.h, self._err)
        return sc.value, g.value, k.value

    def set_deterministic(self, on: bool = True):
        """Deterministic-gradient mode (rlgpu_learner_set_deterministic): dW / db summed in a fixed order instead of with fp32 atomics."""
        _chk(self.lib.rlgpu_learner_set_deterministic(self.h, 1 if on else 0), self.h, self._err)

    def set_lr(self, policy_lr, critic_lr):
        _chk(self.lib.rlgpu_learner_set_lr(self.h, policy_lr, critic_lr), self.h, self._err)

    def sync(self):
        _chk(self.lib.rlgpu_learner_sync(self.h), self.h, self._err)

    def timing_total(self, reset=True):
        """(ms, flops, calls) accumulated over the ppo_minibatch regions since the last reset."""
        ms, fl, n = C.c_float(), C.c_double(), C.c_int()
        _chk(self.lib.rlgpu_learner_timing_total(self.h, C.byref(ms), C.byref(fl), C.byref(n), 1 if reset else 0), self.h, self._err)
        return ms.value, fl.value, n.value

    def last_gemm(self):
        ms, fl = C.c_float(), C.c_double()
        _chk(self.lib.rlgpu_learner_last_gemm(self.h, C.byref(ms), C.byref(fl)), self.h, self._err)
        return ms.value, fl.value


class _DevArray:
    """Minimal __cuda_array_interface__ holder so torch can alias a raw device pointer without copying."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2, "strides": None}


def _wrap_device_f32(ptr, n, device):
    return torch.as_tensor(_DevArray(ptr, n), device=torch.device("cuda", device))
