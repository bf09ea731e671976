"""Multi-GPU glue: one process per GPU, envs sharded across ranks, ONE gradient all-reduce per optimizer step.

The reference is single-process (threads over one learner, PUB/Learner.cpp:436-606); this is what a data-parallel run of it
needs and nothing more.  Everything here works on any torch.distributed backend: `nccl` (= RCCL over xGMI) on the GPUs,
`gloo` in the CPU tests (tests/test_multi_rank_cpu.py).
"""
import os

import torch


def env_ranks():
    """(rank, local_rank, world) as torchrun exports them."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init_process_group(backend="nccl"):
    """Join the job described by the torchrun environment; no-op for a single process."""
    import torch.distributed as dist
    rank, local_rank, world = env_ranks()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this host driver
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def shard_seed(base_seed, rank):
    """Env-shard RNG stream of a rank: disjoint Philox keys, rank 0 reproduces the single-GPU run."""
    return (int(base_seed) + 1000 * int(rank)) & 0xffffffff


def allreduce_gradients(grad, world):
    """Sum the flat [policy | critic] gradient over ranks in place; returns the scale (1/world) that
    rlgpu_clip_adam_step applies BEFORE clipping, so the clip sees the global mean gradient like a single learner would."""
    if world > 1:
        import torch.distributed as dist
        dist.all_reduce(grad)
    return 1.0 / world


def share_from_rank0(t, world):
    """Rank 0's tensor on every rank (the <=150 returns that feed the shared Welford statistic, Learner.cpp:679-682)."""
    if world > 1:
        import torch.distributed as dist
        t = t.clone()
        dist.broadcast(t, src=0)
    return t


def max_over_ranks(value, world, device=None):
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return float(value)


def sum_over_ranks(value, world, device=None):
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([float(value)], dtype=torch.float64, device=device)
        dist.all_reduce(t)
        return float(t.item())
    return float(value)


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
