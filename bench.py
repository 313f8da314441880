#!/usr/bin/env python3
"""Benchmark of the fused plane-sweep + DPV hot path (BASELINE.json metric: depth-volumes/sec).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the fused sweep+DPV kernel over one batch of synthetic inputs already
resident in HBM.  Workload = BASELINE.json configs[1]: mono eval, B=4 volumes per GPU, V=1,
C=67, D=64, sweep resolution 256x512, outputs log-DPV [B,D,H,W] + depth [B,H,W].  With N GPUs
every rank owns its own B=4 batch (weak scaling, no data-path collective); per-rank metrics
are all-gathered once at the end (RCCL).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402

import pdepth_amd  # noqa: E402,F401
from pdepth_amd import dist as pdist  # noqa: E402
from pdepth_amd import ops, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_volume(C, V, D, H, W):
    """fused sweep+DPV reads ref + V src maps once, writes log-DPV + depth."""
    return 4 * H * W * (C * (1 + V) + D + 1)


def cpu_baseline(cfg, budget_s=12.0):
    """Oracle (CPU restatement of the reference) timed on the host cores: bounded sample."""
    from oracle import ref_cpu as O
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    n, t_total = 0, 0.0
    it = synth.make_item(2000, **cfg)
    K = it["K"]
    args = (it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], K, it["rays"],
            K.numpy()[0, 2], K.numpy()[1, 2], 10.0)
    O.sweep_dpv(*args)  # warm-up (allocator, thread pool)
    while t_total < budget_s and n < 16:
        t0 = time.perf_counter()
        O.sweep_dpv(*args)
        t_total += time.perf_counter() - t0
        n += 1
    return {"value": n / t_total, "unit": "depth-volumes/s", "cores": cores, "kind": "port",
            "sample": f"{n} volume(s) of the same workload (1 item, V={cfg['V']}, C={cfg['C']}, D={cfg['D']}, "
                      f"{cfg['H']}x{cfg['W']}) through oracle/ref_cpu.py, torch {torch.__version__} CPU, "
                      f"{cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="volumes per GPU")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--planes", type=int, default=64)
    ap.add_argument("--views", type=int, default=1)
    ap.add_argument("--pose", default="mono", choices=["mono", "stereo"])
    ap.add_argument("--algo", default="auto", choices=["auto", "direct"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank, world, local_rank = pdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    assert world == a.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {a.gpus}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = dict(C=67, D=a.planes, H=a.height, W=a.width, V=a.views, pose=a.pose)
    lo, hi = pdist.shard_range(a.batch * world, rank, world)  # this rank's items of the global batch
    b = synth.make_batch(2, hi - lo, first_item=lo, **cfg)
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], dev)

    def step():
        return ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0,
                             algo=a.algo, want_cost=False, want_logp=True, want_depth=True)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    pdist.barrier()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(a.steps):
        out = step()
        ev[i + 1].record()
    torch.cuda.synchronize(dev)
    pdist.barrier()
    wall = time.perf_counter() - t0
    wall = pdist.max_over_ranks(wall, dev)

    kern_ms = sum(ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)) / a.steps  # HIP events, launch stream
    depth = out[2]
    metrics = torch.tensor([hi - lo, kern_ms, float(depth.mean()), float(torch.isfinite(depth).all())],
                           dtype=torch.float32, device=dev)
    allm = pdist.gather_metrics(metrics).cpu()

    if rank == 0:
        vols = a.batch * world * a.steps
        bytes_per_launch = algorithmic_bytes_per_volume(cfg["C"], cfg["V"], cfg["D"], cfg["H"], cfg["W"]) * (hi - lo)
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        tj = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                traffic = json.load(open(tj)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "depth-volumes/sec (D=64, 256x512)", "value": vols / wall, "unit": "depth-volumes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: default_mono eval, fused sweep+DPV, B={a.batch}/GPU, "
                                   f"V={cfg['V']}, C={cfg['C']}, D={cfg['D']}, {cfg['H']}x{cfg['W']}, pose={a.pose}, "
                                   f"algo={a.algo}", "global_batch": a.batch * world, "parallelism": f"dp{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "fused sweep+DPV = pack_c4_kernel (source re-layout pre-pass) + sweep_tiled_kernel", "kernel_ms": kern_ms,
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         # secondary figure of SURVEY 8(d): flops of the direct formulation, 11*D*h*w*V*C per volume,
                         # against the dense fp32 vector peak (157.3 TFLOP/s); the band mode executes fewer
                         "algorithmic_tflops": 11.0 * cfg["D"] * cfg["H"] * cfg["W"] * cfg["V"] * cfg["C"] * (hi - lo)
                                               / (kern_ms * 1e-3) / 1e12,
                         "fp32_valu_peak_tflops": 157.3},
            "per_rank_kernel_ms": [float(x) for x in allm[:, 1]],
            "gather_fallback_tiles": pdepth_amd._native.fallback_tiles(hi - lo, cfg["H"], cfg["W"]),
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    pdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
