"""One-process-per-GPU plumbing for the hot path.

Batch items are independent, so the path shards with NO data-path collective: rank r owns a
contiguous slice of the batch.  The only exchange is one all_gather of a small per-rank
metric vector at the end of a run -- the counterpart of the reference's shared-memory
``shared[workers, 10]`` averaging (train.py:92, trainer/default_trainer.py:276-283).
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """(rank, world, local_rank).  Initialises torch.distributed when WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of n_items owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def gather_metrics(vec: torch.Tensor) -> torch.Tensor:
    """all_gather of a small fp32 vector -> [world, n] (on vec's device)."""
    if not (dist.is_available() and dist.is_initialized()):
        return vec[None].clone()
    out = [torch.empty_like(vec) for _ in range(dist.get_world_size())]
    dist.all_gather(out, vec.contiguous())
    return torch.stack(out)


def max_over_ranks(x: float, device) -> float:
    t = torch.tensor([x], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
