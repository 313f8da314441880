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


def init_from_env(backend=None, force=None):
    """(rank, world, local_rank).  Initialises torch.distributed when WORLD_SIZE > 1 -- or, with force (argument or
    PDEPTH_DIST_FORCE=1), also for a single rank: the way the test suite runs the RCCL calls of an N > 1 job on a one-GPU box."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("PDEPTH_DIST_FORCE") == "1"
    if (world > 1 or force) and not dist.is_initialized():
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


def assemble_bench_line(allm, wall, *, steps, warmup, batch_per_gpu, world, metric, unit, workload,
                        bytes_per_volume, hbm_peak_gbs, dtype="f32", extras=None):
    """Rank-0 aggregation of a benchmark run -> the dict bench.py prints as its ONE JSON line.

    allm [world, >=2]: the gathered per-rank vectors (column 0 = volumes the rank owns per step, column 1 = its average
    kernel time per step in ms, HIP events); wall = max over ranks of the wall time of the `steps` timed steps.
    value = volumes ALL ranks processed / wall (whole-job throughput); roofline.achieved is per launch, from rank 0's
    kernel time and its own volumes.  Pure function of its arguments (the gloo test drives it with two ranks).
    """
    allm = allm.detach().cpu().double()
    if allm.shape[0] != world:
        raise ValueError(f"gathered metrics of {allm.shape[0]} ranks, world size {world}")
    per_step = float(allm[:, 0].sum())
    vols = per_step * steps
    kern_ms = float(allm[0, 1])
    bytes_per_launch = bytes_per_volume * float(allm[0, 0])
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    line = {
        "metric": metric, "value": vols / wall, "unit": unit, "n_gpus": int(world), "steps": int(steps),
        "warmup": int(warmup), "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": dtype, "data": "synthetic",
        "config": {"workload": workload, "global_batch": int(per_step), "per_gpu_batch": int(batch_per_gpu),
                   "parallelism": f"dp{int(world)}"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": hbm_peak_gbs, "unit": "GB/s",
                     "frac": achieved / hbm_peak_gbs, "traffic": None, "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": bytes_per_launch},
        "per_rank_kernel_ms": [float(x) for x in allm[:, 1]],
    }
    if extras:
        for k, v in extras.items():
            if k == "roofline":
                line["roofline"].update(v)
            else:
                line[k] = v
    return line
