#!/usr/bin/env python3
"""Secondary timings (not the headline metric): every kernel of the path at the BASELINE shapes.

    python tools/bench_kernels.py            # one GPU, prints one JSON line per case
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import pdepth_amd  # noqa: E402,F401
from pdepth_amd import ops, synth  # noqa: E402


def timeit(fn, steps=20, warmup=5, warm_ms=30.0):
    """Mean GPU time per call (HIP events around `steps` back-to-back calls) after `warmup` calls and at least `warm_ms` of
    the same work: from idle the GPU's clocks take ~20 ms of work to come up (profiles/r03_bench_clock_ramp.txt), and the
    early lines of this file used to be 10 % slow against the same case later in the run."""
    import time
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < warm_ms:
        fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def sweep_case(name, B, C, D, H, W, V, pose, algo="auto", steps=20, peaked=False):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose, peaked=peaked)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ms = timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0,
                                      algo=algo), steps=steps)
    byt = 4 * H * W * (C * (1 + V) + D + 1) * B
    print(json.dumps({"case": name, "kernel": "fused sweep+DPV (%s)" % algo, "ms": ms, "volumes_per_s": B / ms * 1e3,
                      "algorithmic_GBps": byt / ms / 1e6, "hbm_frac": byt / ms / 1e6 / 8000.0,
                      "fallback_tiles": pdepth_amd._native.fallback_tiles(B, H, W)}), flush=True)


def line(case, ms, byt, **kw):
    """One roofline line: algorithmic bytes of the case / measured time against the 8 TB/s HBM peak."""
    rec = {"case": case, "ms": ms, "algorithmic_MB": byt / 1e6, "algorithmic_GBps": byt / ms / 1e6,
           "hbm_frac": byt / ms / 1e6 / 8000.0}
    rec.update(kw)
    print(json.dumps(rec), flush=True)


def main():
    sweep_case("cfg2 mono 256x512 B=4", 4, 67, 64, 256, 512, 1, "mono")
    sweep_case("cfg3 stereo 256x512 B=4 (per-GPU share of B=32)", 4, 67, 64, 256, 512, 1, "stereo")
    # the correlated variant of the synthetic features (src = 0.7 shift(ref) + 0.3 noise: a peaked DPV) next to N(0,1)
    sweep_case("cfg2 mono 256x512 B=4, peaked features", 4, 67, 64, 256, 512, 1, "mono", peaked=True)
    sweep_case("cfg3 stereo 256x512 B=4, peaked features", 4, 67, 64, 256, 512, 1, "stereo", peaked=True)
    sweep_case("cfg1/2 model-real 64x128 B=4", 4, 67, 64, 64, 128, 1, "mono", steps=50)
    sweep_case("cfg5 D=128 512x1024 V=4 B=2 (per-GPU share of B=16)", 2, 67, 128, 512, 1024, 4, "mono", steps=5)
    sweep_case("cfg2 mono 256x512 B=4 gather kernel", 4, 67, 64, 256, 512, 1, "mono", algo="direct", steps=5)
    # the implementations behind "auto", forced (A/B): the correlation-form kernel, one-tile and two-tile builds of the tiled kernel
    for algo in ("tiled1", "tiled2"):
        sweep_case("cfg2 mono 256x512 B=4", 4, 67, 64, 256, 512, 1, "mono", algo=algo)
        sweep_case("cfg3 stereo 256x512 B=4", 4, 67, 64, 256, 512, 1, "stereo", algo=algo)
        sweep_case("model-real 64x128 B=4", 4, 67, 64, 64, 128, 1, "mono", algo=algo, steps=50)
    sweep_case("cfg5 D=128 512x1024 V=4 B=2", 2, 67, 128, 512, 1024, 4, "mono", algo="tiled1", steps=5)
    # BASELINE config 1 as the reference runs it (train.py:64-73: eval is B = 1; models.py:518: the sweep at 1/4 resolution):
    # one frame pair, 64x128 (256x512 image) and 64x96 (the default 256x384 crop).  Wall time of back-to-back calls here;
    # the GPU time of the same calls (rocprofv3 kernel trace, launches per call) is in profiles/ (tools/prof_small.sh).
    for (H, W) in ((64, 128), (64, 96)):
        for algo in ("auto", "tiled1"):
            sweep_case("cfg1 one frame pair B=1 %dx%d" % (H, W), 1, 67, 64, H, W, 1, "mono", algo=algo, steps=100)
    # packed-source entry (pdepth_pack_source_f32 once, pdepth_sweep_dpv_packed_f32 per step) and the pre-pass alone
    b = synth.make_batch(2, 4, C=67, D=64, H=256, W=512, V=1, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ps = ops.pack_source(d["src"], 64)
    ms = timeit(lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0))
    line("cfg2 mono 256x512 B=4, packed-source entry", ms, 4 * 256 * 512 * (67 * 2 + 64 + 1) * 4, volumes_per_s=4 / ms * 1e3)
    ms = timeit(lambda: ops.pack_source(d["src"], 64))
    line("pack_source B=4 V=1 C=67 256x512 (read NCHW, write 17+2 float4 planes)", ms, 4 * 256 * 512 * 4 * (67 + 19 * 4))
    # DPV reduction alone at full resolution (decoder output, models.py:351 + default_trainer.py:233)
    for (B, D, H, W) in ((4, 64, 256, 512), (4, 64, 64, 128)):
        x = torch.randn(B, D, H, W, device="cuda")
        y = torch.randn(B, D, H, W, device="cuda")
        dc = ops.d_candi_tensor(synth.powerf(5, 40, D, 1.0), "cuda")
        tag = "B=%d D=%d %dx%d" % (B, D, H, W)
        ms = timeit(lambda: ops.dpv_reduce(x, dc), steps=50)
        line("dpv_reduce " + tag, ms, 4 * H * W * (2 * D + 1) * B)
        ms = timeit(lambda: ops.dpv_expect(x, dc, BV_log=True), steps=50)
        line("dpv_expect " + tag, ms, 4 * H * W * (D + 1) * B)
        ms = timeit(lambda: ops.dpv_moments(x, dc, BV_log=True), steps=50)
        line("dpv_moments " + tag, ms, 4 * H * W * (D + 2) * B)
        ms = timeit(lambda: ops.dpv_reduce_ex(x, dc, want_logp=True, want_depth=True, want_var=True, want_quarter=True), steps=50)
        line("dpv_reduce_ex logp+depth+variance+quarter " + tag, ms, 4 * H * W * (2 * D + 2) * B + 4 * (H // 4) * (W // 4) * D * B)
        ms = timeit(lambda: ops.dpv_reduce_ex(x, dc, addend=y, want_logp=True, want_prob=True), steps=50)
        line("dpv_reduce_ex feedback update (logits+addend -> logp, prob) " + tag, ms, 4 * H * W * 4 * D * B)
        lp = torch.log_softmax(x, dim=1)
        mk = (torch.rand(B, 1, H, W, device="cuda") > 0.6).float()
        dm = (torch.rand(B, H, W, device="cuda") * 30 + 6) * mk[:, 0]
        ms = timeit(lambda: ops.dpv_fuse(lp, dm, mk, dc, 0.3), steps=50)
        line("dpv_fuse (fused + log) " + tag, ms, 4 * H * W * (3 * D + 2) * B)
        intr = torch.tensor([[0.58 * W, 0, W / 2.0, 0, 0.58 * W, H / 2.0, 0, 0, 1]], device="cuda").repeat(B, 1).reshape(B, 3, 3)
        ms = timeit(lambda: ops.ufield(lp, dc, intr, None, BV_log=True), steps=50)
        line("ufield (expectation + band mask + column collapse) " + tag, ms, 4 * H * W * (2 * D + 2) * B,
             note="two reads of the volume: the mask needs the depth map of the whole column first")
    # warp_feature, feedback mode: [B, V=2, 64, 64, 128]
    it = synth.make_batch(4, 4, C=64, D=64, H=64, W=128, V=2, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in it.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ms = timeit(lambda: ops.warp_feature(d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc), steps=50)
    line("warp_feature B=4 V=2 D=64 64x128", ms, 8 * 64 * 128 * 2 * 64 * 4)
    ms = timeit(lambda: ops.sample_coords(d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 64, 128), steps=50)
    line("sample_coords B=4 V=2 D=64 64x128", ms, 8 * 64 * 128 * 2 * 64 * 4 + 12 * 64 * 128 * 4)
    # correlation op (PWC-Lite shapes: C=32..128 at 1/8..1/64 resolution; here B=4, C=64, 64x128, 9x9 displacements)
    x1 = torch.randn(4, 64, 64, 128, device="cuda")
    x2 = torch.randn(4, 64, 64, 128, device="cuda")
    go = torch.randn(4, 81, 64, 128, device="cuda")
    ms = timeit(lambda: pdepth_amd._native.correlation_forward(x1, x2, 4, 1, 4, 1, 1, 1), steps=50)
    line("correlation forward B=4 C=64 64x128 r=4", ms, 4 * 64 * 128 * 4 * (2 * 64 + 81),
         lds_tile_flops=2.0 * 81 * 64 * 64 * 128 * 4, tflops=2.0 * 81 * 64 * 64 * 128 * 4 / ms / 1e9)
    ms = timeit(lambda: pdepth_amd._native.correlation_backward(x1, x2, go, 4, 1, 4, 1, 1, 1), steps=50)
    line("correlation backward (both gradients) B=4 C=64 64x128 r=4", ms, 4 * 64 * 128 * 4 * (4 * 64 + 81),
         tflops=4.0 * 81 * 64 * 64 * 128 * 4 / ms / 1e9)
    from pdepth_amd.utils import inverse_warp as iw
    img = torch.randn(4, 3, 256, 512, device="cuda")
    dep = torch.rand(4, 256, 512, device="cuda") * 30 + 5
    pose = torch.eye(4, device="cuda").repeat(4, 1, 1)
    K = torch.tensor([[300.0, 0, 256], [0, 300.0, 128], [0, 0, 1]], device="cuda").repeat(4, 1, 1)
    ms = timeit(lambda: iw.inverse_warp(img, dep, pose, K), steps=50)
    line("inverse_warp B=4 C=3 256x512 (host wrapper included)", ms, 4 * 256 * 512 * 4 * (3 + 1 + 3) + 4 * 256 * 512)


def model_cases():
    """BASELINE config 4 (mono_feedback, 5 chained frames, B=1, 256x512 image -> 64x128 sweep) and the
    plain eval model: whole-model time per frame (MIOpen convs included) next to the hot-path kernels."""
    import time
    from pdepth_amd import harness
    from pdepth_amd.models import get_model
    for nmode, packed in (("default", True), ("default", False), ("default_feedback", True), ("default_feedback", False)):
        model = get_model(synth.default_cfg(nmode), 0)
        synth.seed_weights(model, seed=8)
        model = model.cuda().eval()
        model.packed_epilogue = packed   # True: encoder epilogue kernel + packed sweep entry; False: cat + avg_pool2d + plain entry
        frames = [harness.move_input(synth.make_model_input(4000 + i, B=1, V=1, H=256, W=512, D=64, pose="mono"), "cuda")
                  for i in range(5)]
        harness.eval_trajectory(model, frames)  # warm-up (MIOpen find)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            harness.eval_trajectory(model, frames)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 15 * 1e3
        print(json.dumps({"case": "cfg4-style whole model nmode=%s, 5-frame trajectory, B=1, 256x512 image" % nmode,
                          "encoder_epilogue": "pack_views kernel + packed sweep entry" if packed else "torch.cat + avg_pool2d + plain entry",
                          "ms_per_frame": ms, "frames_per_s": 1e3 / ms}), flush=True)
    # the head alone (PacknetHead: sweep -> log-softmax -> E[d]) from the encoder output, both ways, at the model's own
    # 1/4 resolution and at the benchmark's 256x512
    import torch.nn.functional as F
    for (B, h, w) in ((1, 64, 128), (4, 256, 512)):
        g = torch.Generator().manual_seed(3)
        feat = torch.randn(B * 2, 64, h, w, generator=g).cuda()
        rgb = torch.rand(B * 2, 3, 4 * h, 4 * w, generator=g).cuda()
        it = synth.make_batch(2, B, C=67, D=64, H=h, W=w, V=1, pose="mono")
        d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in it.items()}
        dc = ops.d_candi_tensor(d["d_candi"], "cuda")
        cam = (d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)

        def old():
            both = torch.cat((feat, F.avg_pool2d(rgb, 4)), dim=1).view(B, 2, 67, h, w)
            return ops.sweep_dpv(both[:, -1], both[:, :-1], *cam)

        def new():
            ps, ref = ops.pack_views(feat, rgb, 2, 64)
            return ops.sweep_dpv(ref, ps, *cam)
        for name, fn in (("torch.cat + avg_pool2d + pdepth_sweep_dpv_f32 (pre-pass inside)", old), ("pdepth_pack_views_f32 + pdepth_sweep_dpv_packed_f32", new)):
            ms = timeit(fn, steps=50)
            print(json.dumps({"case": "head from encoder output, B=%d %dx%d" % (B, h, w), "path": name, "ms": ms}), flush=True)


if __name__ == "__main__":
    main()
    model_cases()
