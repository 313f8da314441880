#!/usr/bin/env python3
"""Benchmark of the fused plane-sweep + DPV hot path (BASELINE.json metric: depth-volumes/sec).

    python bench.py --gpus N --steps K --warmup W          # N > 1 without a launcher: starts N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the fused sweep+DPV entry point (pdepth_sweep_dpv_f32, NCHW features in, log-DPV + depth out)
over one batch of synthetic inputs already resident in HBM.  Workload = BASELINE.json configs[1]: mono eval, B=4
volumes per GPU, V=1, C=67, D=64, sweep resolution 256x512.  With N GPUs every rank owns its own B=4 batch (weak
scaling, no data-path collective); per-rank metrics are all-gathered once at the end (RCCL).  Prints ONE JSON line on
rank 0.  Next to the headline it reports the same step through the packed-source entry
(pdepth_sweep_dpv_packed_f32: features already in the kernels' staging layout) as `packed_entry`, and SURVEY 8(d)'s
secondary figures: `model_real` (the shape the reference's model sweeps a 256x512 frame at, 64x128, B = 1 and 4, both entries)
and `peaked` (the headline workload on the correlated feature variant).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
# wave-instructions / s: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 clocks at 2.4 GHz (16 lanes x 4 passes; the 157 TFLOP/s
# fp32 peak = this rate x 64 lanes x 2 flop x 2 packed components).  Rounds 1-3 divided by 2 clocks and under-reported valu_frac by 2x:
# SQ_ACTIVE_INST_VALU of the same runs (one quad-cycle per VALU instruction) says 4.
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4


def algorithmic_bytes_per_volume(C, V, D, H, W):
    """fused sweep+DPV reads ref + V src maps once, writes log-DPV + depth."""
    return 4 * H * W * (C * (1 + V) + D + 1)


def cpu_baseline(cfg, budget_s=45.0, gpu_depth0=None):
    """Oracle (CPU restatement of the reference) timed on the host cores: a bounded sample -- ONE volume of the same
    workload per thread count -- at all, 64, 32, 8 and 1 threads (a leg is skipped once the budget is spent; the
    1-thread leg alone takes ~20 s).  value = the best of them, value_1t = one thread.  The volume is the first item of
    the timed batch: its depth map doubles as a check of what the GPU produced for it (`gpu_depth0`, a CPU tensor)."""
    import torch
    from pdepth_amd import synth
    from oracle import ref_cpu as O
    cores = os.cpu_count() or 1
    it = synth.make_item(2000, **cfg)
    K = it["K"]
    args = (it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], K, it["rays"], K.numpy()[0, 2],
            K.numpy()[1, 2], 10.0)
    by_threads, t_used = {}, 0.0
    for n in sorted({1, 8, 32, 64, cores}, reverse=True):
        if n > cores or t_used > budget_s:
            continue
        torch.set_num_threads(n)
        t0 = time.perf_counter()
        odepth = O.sweep_dpv(*args)[2]
        dt = time.perf_counter() - t0
        t_used += dt
        by_threads[str(n)] = 1.0 / dt
    best = max(by_threads, key=by_threads.get)
    check = None
    if gpu_depth0 is not None and by_threads:
        check = float((odepth.reshape(gpu_depth0.shape) - gpu_depth0).abs().max())
    return {"max_abs_depth_diff_gpu_vs_port_item0": check, "value": by_threads[best], "unit": "depth-volumes/s", "cores": int(best), "threads_best": int(best),
            "value_1t": by_threads.get("1"), "by_threads": by_threads, "host_cores": cores, "kind": "port",
            "sample": f"1 volume per thread count of the same workload (1 item, V={cfg['V']}, C={cfg['C']}, "
                      f"D={cfg['D']}, {cfg['H']}x{cfg['W']}) through oracle/ref_cpu.py, torch CPU; "
                      f"{t_used:.0f} s of CPU work in total"}


def self_launch(a):
    """--gpus N without a launcher: start N fresh ranks (one per GPU) BEFORE this process touches the GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


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
    ap.add_argument("--algo", default="auto", choices=["auto", "direct", "dist", "corr", "tiled1", "tiled2", "cells", "mfma"],
                    help="corr / cells / mfma: lab builds of the library only (make LAB=1)")
    ap.add_argument("--peaked", action="store_true", help="SURVEY 8(d)'s correlated feature variant (src = 0.7 shift(ref) + 0.3 noise: a peaked DPV)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-start measurement in front of the headline")
    ap.add_argument("--preheat-ms", type=float, default=60.0, help="untimed headline steps in front of the W warm-up steps (clock ramp), ms of wall time")
    ap.add_argument("--no-secondary", action="store_true", help="skip the model_real / peaked measurements in front of the headline")
    ap.add_argument("--no-workloads", action="store_true", help="skip the `workloads` object (BASELINE configs[2] and [4] at their per-GPU shard, behind the headline)")
    ap.add_argument("--config", default=None, help="an experiment file in the reference's JSON schema (configs/*.json): "
                    "planes, depth range, sigma, channels and the pose family come from it; the sweep resolution stays "
                    "--height x --width (BASELINE quotes the metric at 256x512)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))   # non-zero if any rank fails

    import torch
    import pdepth_amd
    from pdepth_amd import dist as pdist
    from pdepth_amd import ops, synth

    rank, world, local_rank = pdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    assert world == a.gpus, f"WORLD_SIZE={world} but --gpus {a.gpus}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    cfg = dict(C=67, D=a.planes, H=a.height, W=a.width, V=a.views, pose=a.pose)
    if a.peaked:
        cfg["peaked"] = True
    sigma, d_candi_cfg = 10.0, None
    if a.config:   # the reference's experiment file names the workload (BASELINE configs are named by those files)
        wl = synth.sweep_workload(synth.cfg_from_json(a.config))
        cfg.update(C=wl["C"], D=wl["D"], pose=wl["pose"])
        sigma, d_candi_cfg = wl["sigma"], wl["d_candi"]
    lo, hi = pdist.shard_range(a.batch * world, rank, world)  # this rank's items of the global batch
    b = synth.make_batch(2, hi - lo, first_item=lo, **cfg)
    if d_candi_cfg is not None:
        b["d_candi"] = d_candi_cfg
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], dev)

    def step(src):
        return ops.sweep_dpv(d["ref"], src, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, sigma,
                             algo=a.algo, want_cost=False, want_logp=True, want_depth=True)

    def timed(src):
        for _ in range(a.warmup):
            step(src)
        torch.cuda.synchronize(dev)
        pdist.barrier()
        # HIP events on the launch stream (torch's current stream IS the stream the C ABI launches on) around the K steps.
        # One pair for the whole region: an event between two steps is a packet of its own on the queue, and with one after
        # every step the K steps ran 2 % (NCHW entry) to 10 % (packed entry: one kernel per step) slower than the same K
        # calls back to back (profiles/r05_ab/bench_events_per_step.txt); PDEPTH_BENCH_TRACE=1 brings the per-step events
        # back (diagnostics: clock ramp, stragglers).
        trace = bool(os.environ.get("PDEPTH_BENCH_TRACE"))
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1 if trace else 2)]
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(a.steps):
            out = step(src)
            if trace:
                ev[i + 1].record()
        if not trace:
            ev[1].record()
        torch.cuda.synchronize(dev)
        pdist.barrier()
        wall = pdist.max_over_ranks(time.perf_counter() - t0, dev)
        kern_ms = ev[0].elapsed_time(ev[-1]) / a.steps
        if trace:
            print("per-step ms: " + " ".join("%.4f" % ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)), file=sys.stderr)
        return out, wall, kern_ms

    # Order of the GPU work of this process, all of it reported in the line:
    #   1. `cold_start`: the headline protocol -- W warm-up steps, K timed steps -- as the FIRST GPU work of the process.  From idle
    #      the GPU's clocks take ~20 ms of work to come up, more than the driver's 5 warm-up steps: these K steps run on rising clocks
    #      (PDEPTH_BENCH_TRACE=1 prints them: 0.59 ms falling to 0.47 in round 3).
    #   2. `preflight`: the gather kernel -- the reference's op order, pinned to the oracle by the tests -- on the very batch that is
    #      timed, as a cross-check of the headline output.
    #   3. `model_real`: SURVEY 8(d)'s 64x128 figures.
    #   4. --preheat-ms of the headline step, untimed; then the headline: W warm-up steps, K timed steps, at sustained clocks.
    #      `preheat_ms` = the wall time of 1-4 in front of it.
    #   5. `packed_entry`: the same step on features already in the kernels' staging layout (W + K steps), right behind; then
    #      `peaked`: the headline call on SURVEY 8(d)'s correlated features (W + K steps).
    t_pre0 = time.perf_counter()
    cold = None
    if not a.no_cold:
        _, _, cold_ms = timed(d["src"])
        cold = {"ms_per_step": cold_ms, "what": "the same W warm-up + K timed steps as the first GPU work of the process (clocks still rising)"}
    depth_gather = None
    try:
        depth_gather = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, sigma,
                                     algo="direct", want_cost=False, want_logp=False, want_depth=True)[2]
    except RuntimeError:
        pass
    # SURVEY 8(d), "Ambiguity resolved": the 64x128 model-real number is always reported next to the 256x512 one, and both the
    # N(0,1) and the peaked feature variants.  GPU time per call from HIP events over 50 back-to-back calls (the launch gaps of
    # the call are inside); launches per call: NCHW entry = statistics + pack + sweep + the gather kernel for routed items (its
    # blocks leave at once when none is), packed entry = the sweep.
    secondary = {}
    if not a.no_secondary and a.algo in ("auto", "dist", "corr") and rank == 0 and not a.config:
        def small(Bs, entry):
            bs = synth.make_batch(2, Bs, C=cfg["C"], D=cfg["D"], H=64, W=128, V=cfg["V"], pose=a.pose)
            ds = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in bs.items()}
            src = ops.pack_source(ds["src"], cfg["D"], a.algo) if entry == "packed" else ds["src"]
            f = lambda: ops.sweep_dpv(ds["ref"], src, ds["K"], ds["R"], ds["t"], ds["rays"], ds["cxcy"], dc, sigma, algo=a.algo)
            for _ in range(10):
                f()
            torch.cuda.synchronize(dev)
            us, wall_us = float("inf"), float("inf")
            for _ in range(3):   # (the best of three rounds of 50 calls: one round in ten catches a stall of the box of 100 us and more)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                for _ in range(50):
                    f()
                e1.record()
                torch.cuda.synchronize(dev)
                us = min(us, e0.elapsed_time(e1) / 50 * 1e3)
                wall_us = min(wall_us, (time.perf_counter() - t0) / 50 * 1e6)
            by = algorithmic_bytes_per_volume(cfg["C"], cfg["V"], cfg["D"], 64, 128) * Bs
            return {"us_per_call": us, "wall_us_per_call": wall_us, "launches": 1 if entry == "packed" else 4,
                    "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "volumes_per_s": Bs / (us * 1e-6)}
        try:
            secondary["model_real"] = {"shape": "C=%d D=%d 64x128 V=%d pose=%s (the reference sweeps a 256x512 frame at 1/4 resolution: "
                                                "models/models.py:518)" % (cfg["C"], cfg["D"], cfg["V"], a.pose),
                                       "what": "us_per_call = HIP events around 50 back-to-back calls of the Python binding, best of three rounds (at B=1 the host's call, ~26 us, is "
                                               "longer than the kernel: profiles/r06_small_sweeps.rocprofv3.txt has the kernels' own durations); "
                                               "frac = algorithmic bytes / us_per_call / 8 TB/s",
                                       "B1_nchw": small(1, "nchw"), "B1_packed": small(1, "packed"),
                                       "B4_nchw": small(4, "nchw"), "B4_packed": small(4, "packed")}
        except RuntimeError as e:
            secondary["error"] = str(e)
    dp = None
    if "model_real" in secondary and not a.peaked:   # the peaked variant's batch, made on the host BEFORE the preheat (the GPU idles meanwhile)
        bp = synth.make_batch(2, hi - lo, first_item=lo, **dict(cfg, peaked=True))
        dp = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in bp.items()}
        del bp
    # sustained clocks: from idle (and after the small / host-bound calls above) the GPU's clocks take ~50 steps of this size to
    # settle (PDEPTH_BENCH_TRACE=1 shows the steps falling by 10 %; profiles/r05_ab/bench_clock_ramp.txt); the headline's own W warm-up
    # steps are 2.5 ms.  60 ms of the headline step, untimed, declared in the line (`preheat`); `cold_start` is the number without it.
    pdist.barrier()   # (N > 1: rank 0 ran the secondary measurements above; every rank starts its preheat now, not a second before)
    t_h0 = time.perf_counter()
    while (time.perf_counter() - t_h0) * 1e3 < a.preheat_ms:
        for _ in range(8):
            step(d["src"])
        torch.cuda.synchronize(dev)
    preheat_ms = (time.perf_counter() - t_pre0) * 1e3

    out, wall, kern_ms = timed(d["src"])
    packed_entry, out_p, kern_p = None, None, None
    if a.algo in ("auto", "dist", "corr"):   # secondary: the same step on features already in the kernels' staging layout, right behind
        try:
            ps = ops.pack_source(d["src"], cfg["D"], a.algo)
            out_p, _, kern_p = timed(ps)
            del ps
        except RuntimeError as e:
            packed_entry = {"error": str(e)}
    if dp is not None:   # the peaked variant of the headline call, at sustained clocks like the headline
        try:
            fp = lambda: ops.sweep_dpv(dp["ref"], dp["src"], dp["K"], dp["R"], dp["t"], dp["rays"], dp["cxcy"], dc, sigma, algo=a.algo)
            for _ in range(a.warmup):
                fp()
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                fp()
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / a.steps
            secondary["peaked"] = {"ms_per_step": ms, "value": (hi - lo) / (ms * 1e-3),
                                   "what": "the headline call on SURVEY 8(d)'s correlated features (src = 0.7 shift(ref) + 0.3 noise)"}
            del dp
        except RuntimeError as e:
            secondary["error"] = str(e)
    # BASELINE configs[2] (default_stereo, batch 32 over 8 GPUs) and configs[4] (D = 128, 512x1024, 4 views, batch 16 over 8
    # GPUs) at the shard ONE GPU gets (B = 4 / B = 2): the driver runs one command, so their numbers ride in this line.  NCHW
    # entry (statistics + pack + sweep), HIP events around W + K calls; each checked against the gather kernel (the
    # reference's op order) on the same batch.
    workloads = {}
    if not a.no_workloads and not a.no_secondary and a.algo in ("auto", "dist") and rank == 0 and not a.config:
        def shard(name, Bs, D, H, W, V, pose, steps, offset=0.0, smooth=0):
            bs = synth.make_batch(2, Bs, C=cfg["C"], D=D, H=H, W=W, V=V, pose=pose)
            if smooth:   # spatially smooth features (box filter, unit variance again): what the trend guard looks at
                blur = lambda x: torch.nn.functional.avg_pool2d(x, smooth, 1, smooth // 2, count_include_pad=False)
                r, sr = blur(bs["ref"]), blur(bs["src"].flatten(0, 1)).view_as(bs["src"])
                bs["ref"], bs["src"] = r / r.std(), sr / sr.std()
            if offset:   # per-channel offsets, the same in every view: costs of hundreds where a tap leaves the image
                mu = (torch.rand(cfg["C"], generator=torch.Generator().manual_seed(5)) * 2 - 1) * offset
                bs["ref"] += mu[None, :, None, None]
                bs["src"] += mu[None, None, :, None, None]
            ds = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in bs.items()}
            dcs = ops.d_candi_tensor(ds["d_candi"], dev)
            f = lambda algo: ops.sweep_dpv(ds["ref"], ds["src"], ds["K"], ds["R"], ds["t"], ds["rays"], ds["cxcy"], dcs, sigma, algo=algo)
            for _ in range(a.warmup):
                o = f(a.algo)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                o = f(a.algo)
            e1.record()
            torch.cuda.synchronize(dev)
            ms = e0.elapsed_time(e1) / steps
            direct_passes = pdepth_amd._native.fallback_tiles(Bs, H, W)
            impl_s = a.algo if a.algo != "auto" else pdepth_amd._native.selected_kernel(Bs, V, cfg["C"], D, H, W)
            ref_depth = f("direct")[2]
            by = algorithmic_bytes_per_volume(cfg["C"], V, D, H, W) * Bs
            del ds
            return {"workload": name, "B": Bs, "ms_per_call": ms, "volumes_per_s": Bs / (ms * 1e-3), "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "sweep_kernel": impl_s, "direct_passes": direct_passes,
                    "max_abs_depth_diff_vs_gather": float((o[2] - ref_depth).abs().max())}
        try:
            workloads["cfg3_shard"] = shard("BASELINE configs[2] at one GPU's shard: default_stereo eval, D=64, 256x512, V=1, B=4 (of 32 over 8 GPUs)",
                                            4, 64, 256, 512, 1, "stereo", a.steps)
            workloads["cfg5_share"] = shard("BASELINE configs[4] at one GPU's share: D=128, 512x1024, 4 source views, B=2 (of 16 over 8 GPUs)",
                                            2, 128, 512, 1024, 4, "mono", max(3, a.steps // 4))
            # the perf cliff of the default selector, in the open: features whose channel means are 6 sigma off zero make an
            # item ill-conditioned by the kernel's measure (csrc/sweep_dist.hip: "Conditioning"); every item is then the gather kernel's
            workloads["cfg2_smooth"] = shard("the headline shape with spatially smooth features (33x33 box filter of N(0,1), unit variance: var / lag-16 spread -- "
                                             "the trend guard's ratio -- is 2, the energy against sigma is small): the fast form, nothing routed",
                                             hi - lo, cfg["D"], cfg["H"], cfg["W"], cfg["V"], a.pose, max(3, a.steps // 2), smooth=33)
            workloads["cfg2_routed"] = shard("the headline shape with per-channel offsets of up to 6 sigma: every item routed to the gather kernel "
                                             "(direct_passes = pixel blocks routed)", hi - lo, cfg["D"], cfg["H"], cfg["W"], cfg["V"], a.pose,
                                             max(3, a.steps // 4), offset=6.0)
        except RuntimeError as e:
            workloads["error"] = str(e)
    depth = out[2]
    metrics = torch.tensor([hi - lo, kern_ms, float(depth.mean()), float(torch.isfinite(depth).all())],
                           dtype=torch.float32, device=dev)
    allm = pdist.gather_metrics(metrics).cpu()
    fallback = pdepth_amd._native.fallback_tiles(hi - lo, cfg["H"], cfg["W"], gather_flag=2 if a.algo == "cells" else 1)
    # which sweep kernel ran: the selector, or -- ALGO_AUTO -- the family the library picks for this descriptor
    impl = a.algo if a.algo != "auto" else pdepth_amd._native.selected_kernel(hi - lo, cfg["V"], cfg["C"], cfg["D"], cfg["H"], cfg["W"])

    if out_p is not None:
        packed_entry = {"kernel_ms": kern_p, "max_abs_depth_diff_vs_headline": float((out_p[2] - depth).abs().max())}
    preflight = None
    if depth_gather is not None:
        preflight = {"max_abs_depth_diff_vs_gather_kernel": float((depth_gather - depth).abs().max()),
                     "what": "sweep_direct_kernel (reference op order) on the timed batch, run once before the timed region"}

    if rank == 0:
        bpv = algorithmic_bytes_per_volume(cfg["C"], cfg["V"], cfg["D"], cfg["H"], cfg["W"])
        prof = {}
        tj = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            # committed PMC counters (tools/prof.sh): attached only to the workload AND the kernel they were collected on
            try:
                for rec in json.load(open(tj)).get("workloads", []):
                    w = rec.get("workload", {})
                    if (w.get("batch"), w.get("C"), w.get("D"), w.get("H"), w.get("W"), w.get("V"), w.get("pose"), w.get("kernel")) == \
                            (hi - lo, cfg["C"], cfg["D"], cfg["H"], cfg["W"], cfg["V"], cfg["pose"], impl):
                        prof = rec
            except Exception:
                prof = {}
        kname = {"dist": "sweep_dist_kernel (distance form sum w |s - r|^2 - Q, fp16 high / low parts on the matrix pipe)",
                 "corr": "sweep_corr_kernel (correlation form on mean-centred features, matrix pipe)",
                 "mfma": "sweep_mfma_kernel (matrix-pipe kernel)", "tiled": "sweep_tiled_kernel (LDS-tiled band kernel)",
                 "tiled1": "sweep_tiled_kernel, one tile per block", "tiled2": "sweep_tiled_kernel, two tiles per block",
                 "cells": "sweep_cells_fast_kernel + sweep_cells_kernel", "direct": "sweep_direct_kernel (gather)"}.get(impl, str(impl))
        roof = {"traffic": prof.get("hbm_bytes_per_launch"),
                # (PMC counters cannot be collected inside this run: the record is the committed one of this workload and kernel,
                #  with the commit and box it was collected on)
                "traffic_source": prof.get("source"),
                "traffic_collected": prof.get("collected"),
                "kernel": ("fused sweep+DPV call = feature_stats_kernel + pack_dist_kernel (pre-pass: channel statistics, centred fp16 re-layout "
                           "+ neighbour differences) + " + kname + " + sweep_direct_kernel in per-item mode (the gather kernel for the items the "
                           "sweep routes to it: none on this workload, its blocks read one flag and leave)" if impl == "dist" else
                           "fused sweep+DPV call = feature_stats_kernel + pack_c4_kernel (pre-pass: channel means, centred re-layout) + " + kname
                           if impl == "corr" else
                           "fused sweep+DPV call = feature_stats_kernel + pack_c4_kernel (source re-layout pre-pass) + " + kname +
                           " + the gather kernel on flagged tiles"),
                "sweep_kernel": impl,
                # secondary figure of SURVEY 8(d): flops of the direct formulation, 11*D*h*w*V*C per volume, against
                # the dense fp32 vector peak (157.3 TFLOP/s); the band mode executes fewer
                "algorithmic_tflops": 11.0 * cfg["D"] * cfg["H"] * cfg["W"] * cfg["V"] * cfg["C"] * (hi - lo)
                                      / (kern_ms * 1e-3) / 1e12,
                "fp32_valu_peak_tflops": 157.3}
        if prof.get("valu_wave_instr_per_launch"):
            # VALU wave-instructions per launch (PMC SQ_INSTS_VALU, committed profile) / issue peak / kernel time
            roof["valu_frac"] = prof["valu_wave_instr_per_launch"] / (kern_ms * 1e-3) / VALU_ISSUE_PEAK
            roof["valu_wave_instr_per_launch"] = prof["valu_wave_instr_per_launch"]
        extras = {"roofline": roof, "gather_fallback_tiles": fallback,
                  "preheat_ms": preheat_ms, "preheat": "cold_start + preflight + model_real (see those objects) and %.0f ms of the headline step, untimed, ran before the headline's W warm-up steps; packed_entry and peaked right behind the headline" % a.preheat_ms}
        for k in ("model_real", "peaked"):
            if k in secondary:
                extras[k] = secondary[k]
        if workloads:
            extras["workloads"] = workloads
        if "error" in secondary:
            extras["secondary_error"] = secondary["error"]
        if cold is not None:
            cold["value"] = (hi - lo) * world / (cold["ms_per_step"] * 1e-3)
            extras["cold_start"] = cold
        if preflight is not None:
            extras["preflight"] = preflight
        if packed_entry is not None:
            if "kernel_ms" in packed_entry:
                pa = bpv * (hi - lo) / (packed_entry["kernel_ms"] * 1e-3) / 1e9
                packed_entry.update({"achieved": pa, "frac": pa / HBM_PEAK_GBS, "unit": "GB/s",
                                     "what": "pdepth_sweep_dpv_packed_f32: source already packed "
                                             "(pdepth_pack_source_f32 outside the timed region)"})
            extras["packed_entry"] = packed_entry
        line = pdist.assemble_bench_line(
            allm, wall, steps=a.steps, warmup=a.warmup, batch_per_gpu=a.batch, world=world,
            metric="depth-volumes/sec (D=64, 256x512)", unit="depth-volumes/s",
            workload=("BASELINE configs[1]: default_mono eval" if a.pose == "mono" else "BASELINE configs[2] (one GPU's shard of the batch): default_stereo eval") +
                     f", fused sweep+DPV, B={a.batch}/GPU, V={cfg['V']}, "
                     f"C={cfg['C']}, D={cfg['D']}, {cfg['H']}x{cfg['W']}, pose={a.pose}, algo={a.algo}" + (", peaked features" if a.peaked else ""),
            bytes_per_volume=bpv, hbm_peak_gbs=HBM_PEAK_GBS, extras=extras,
            # I/O and every sum in fp32; the default kernel forms the channel contraction from fp16 high / low pairs of the
            # centred features on the matrix pipe (22 bits per feature, products exact, fp32 accumulation: DESIGN.md section 1)
            dtype="f32 (fp16-pair products on MFMA, fp32 accumulate)" if impl == "dist" else "f32")
        if world == 1 and not a.no_cpu_baseline:
            # (with --config the depth candidates / sigma come from the file, the port's sample keeps the defaults: timing only)
            line["cpu_baseline"] = cpu_baseline(cfg, gpu_depth0=None if a.config else depth[0].cpu())
        print(json.dumps(line), flush=True)
    pdist.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
