"""Multi-GPU layer: one process per GPU, graphs sharded contiguously, ONE all-reduce per step.

The reference's only parallelism is instance-level: `Pool.apply_async(batch_sgd, ...)` with the
per-instance step vectors added into global arrays by `batch_sgd_accumulate` under a lock
(train_mp.py:405-424, 634-649).  Graphs are independent, so sweeps need no communication; the
accumulate callback becomes one `all_reduce(SUM)` over a single fused float64 buffer per
optimisation step (RCCL over xGMI on MI355X; `gloo` in the CPU tests).  At ~14 doubles the
collective is pure latency: it is issued once per step, never per sweep.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  Single-process when WORLD_SIZE is absent or 1."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_range(n_items, rank, world):
    """Contiguous split of n_items over `world` ranks; the first
    n_items % world ranks take one extra item.  Returns (start, stop)."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError('bad rank/world')
    base, extra = divmod(int(n_items), world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_reduce_sum_(buf):
    """In-place SUM all-reduce of one fused statistics buffer; a no-op without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def barrier():
    """All ranks meet (no-op without a process group): rank 0 joining the ranks' prediction shards into one file."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
