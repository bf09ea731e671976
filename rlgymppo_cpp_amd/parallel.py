"""Multi-GPU glue: one process per GPU, envs sharded across ranks, ONE gradient all-reduce per optimizer step.

The reference is single-process (threads over one learner, PUB/Learner.cpp:436-606); this is what a data-parallel run of it
needs and nothing more.  The exchange lives in the C-ABI (include/rlgpu.h "multi-GPU": rlgpu_comm_* on RCCL, rendezvous through
the launcher's environment) -- `RcclComm` below only forwards to it.  `GlooComm` runs the same host logic over torch.distributed
for the CPU tests (tests/test_multi_rank_cpu.py); it is not used on a GPU box.
"""
import ctypes as C
import os

import torch


def env_ranks():
    """(rank, local_rank, world) as torchrun exports them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_seed(base_seed, rank):
    """Env-shard RNG stream of a rank: disjoint Philox keys, rank 0 reproduces the single-GPU run."""
    return (int(base_seed) + 1000 * int(rank)) & 0xffffffff


class SoloComm:
    """world size 1."""
    rank, world = 0, 1

    def allreduce_gradients(self, ppo):
        return 1.0

    def share_from_rank0(self, t):
        return t

    def max_over_ranks(self, value, device=None):
        return float(value)

    def sum_over_ranks(self, value, device=None):
        return float(value)

    def barrier(self):
        pass

    def close(self):
        pass


class RcclComm(SoloComm):
    """rlgpu_comm (RCCL over xGMI) from the launcher's environment: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_PORT."""

    def __init__(self):
        from . import _lib
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver
        self.lib = _lib.load()
        self.h = C.c_void_p()
        r, w = C.c_int(), C.c_int()
        rc = self.lib.rlgpu_comm_init_env(C.byref(self.h), C.byref(r), C.byref(w))
        if rc != 0:
            raise _lib.RlgpuError(f"rlgpu_comm_init_env failed ({rc}): {self.lib.rlgpu_comm_last_error(None).decode()}")
        self.rank, self.world = r.value, w.value
        self.device = self.lib.rlgpu_comm_device(self.h)    # LOCAL_RANK: the learner whose gradients this reduces must live there
        self._lib_mod = _lib

    def _chk(self, rc):
        if rc != 0:
            raise self._lib_mod.RlgpuError(f"rlgpu_comm error {rc}: {self.lib.rlgpu_comm_last_error(self.h).decode()}")

    def allreduce_gradients(self, ppo):
        """Sum the flat [policy | critic] gradient over ranks in place on the learner's stream; returns the scale (1 / world) that
        rlgpu_clip_adam_step applies BEFORE clipping, so the clip sees the global mean gradient like a single learner would."""
        self._chk(self.lib.rlgpu_allreduce_grads(ppo.h, self.h))
        return 1.0 / self.world

    def _allreduce(self, t):
        assert t.dtype == torch.float32 and t.is_cuda and t.is_contiguous()
        self._chk(self.lib.rlgpu_comm_allreduce_f32(self.h, C.c_void_p(t.data_ptr()), t.numel(), C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))
        return t

    def share_from_rank0(self, t):
        t = t.clone().contiguous()
        self._chk(self.lib.rlgpu_comm_broadcast(self.h, C.c_void_p(t.data_ptr()), t.numel() * t.element_size(), 0, C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)))
        return t

    def _gather_scalar(self, value, device):
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        t = torch.zeros(self.world, dtype=torch.float32, device=device)
        t[self.rank] = float(value)
        return self._allreduce(t).double().cpu()

    def max_over_ranks(self, value, device=None):
        return float(self._gather_scalar(value, device).max())

    def sum_over_ranks(self, value, device=None):
        return float(self._gather_scalar(value, device).sum())

    def barrier(self):
        dev = torch.device("cuda", torch.cuda.current_device())
        self._allreduce(torch.zeros(1, dtype=torch.float32, device=dev)); torch.cuda.synchronize(dev)

    def close(self):
        if self.h:
            self.lib.rlgpu_comm_destroy(self.h); self.h = C.c_void_p()


class GlooComm(SoloComm):
    """CPU tests only: the same calls over torch.distributed (`gloo`); the gradient is reduced through the tensor view."""

    def __init__(self, backend="gloo"):
        import torch.distributed as dist
        rank, local_rank, world = env_ranks()
        if world > 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend, rank=rank, world_size=world)
        self.rank, self.world, self.dist = rank, world, dist

    def allreduce_tensor(self, grad):
        if self.world > 1:
            self.dist.all_reduce(grad)
        return 1.0 / self.world

    def allreduce_gradients(self, ppo):
        return self.allreduce_tensor(ppo.grad_tensor())

    def share_from_rank0(self, t):
        if self.world > 1:
            t = t.clone(); self.dist.broadcast(t, src=0)
        return t

    def max_over_ranks(self, value, device=None):
        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value, device=None):
        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.world > 1:
            self.dist.all_reduce(t)
        return float(t.item())

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.barrier(); self.dist.destroy_process_group()


def make_comm():
    """The communicator of this process: RCCL through the C-ABI when the launcher started more than one rank, else a no-op."""
    _, local_rank, world = env_ranks()
    if world > 1:
        torch.cuda.set_device(local_rank)
        return RcclComm()
    return SoloComm()
