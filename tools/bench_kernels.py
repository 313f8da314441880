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


def timeit(fn, steps=20, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps


def sweep_case(name, B, C, D, H, W, V, pose, algo="auto", steps=20):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ms = timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0,
                                      algo=algo), steps=steps)
    byt = 4 * H * W * (C * (1 + V) + D + 1) * B
    print(json.dumps({"case": name, "kernel": "fused sweep+DPV (%s)" % algo, "ms": ms, "volumes_per_s": B / ms * 1e3,
                      "algorithmic_GBps": byt / ms / 1e6, "hbm_frac": byt / ms / 1e6 / 8000.0,
                      "fallback_tiles": pdepth_amd._native.fallback_tiles(B, H, W)}), flush=True)


def main():
    sweep_case("cfg2 mono 256x512 B=4", 4, 67, 64, 256, 512, 1, "mono")
    sweep_case("cfg3 stereo 256x512 B=4 (per-GPU share of B=32)", 4, 67, 64, 256, 512, 1, "stereo")
    sweep_case("cfg1/2 model-real 64x128 B=4", 4, 67, 64, 64, 128, 1, "mono", steps=50)
    sweep_case("cfg5 D=128 512x1024 V=4 B=2 (per-GPU share of B=16)", 2, 67, 128, 512, 1024, 4, "mono", steps=5)
    sweep_case("cfg2 mono 256x512 B=4 gather kernel", 4, 67, 64, 256, 512, 1, "mono", algo="direct", steps=5)
    # DPV reduction alone at full resolution (decoder output, models.py:351 + default_trainer.py:233)
    for (B, D, H, W) in ((4, 64, 256, 512), (4, 64, 64, 128)):
        x = torch.randn(B, D, H, W, device="cuda")
        dc = ops.d_candi_tensor(synth.powerf(5, 40, D, 1.0), "cuda")
        ms = timeit(lambda: ops.dpv_reduce(x, dc), steps=50)
        byt = 4 * H * W * (2 * D + 1) * B
        print(json.dumps({"case": "dpv_reduce B=%d D=%d %dx%d" % (B, D, H, W), "ms": ms,
                          "algorithmic_GBps": byt / ms / 1e6, "hbm_frac": byt / ms / 1e6 / 8000.0}), flush=True)
        ms = timeit(lambda: ops.dpv_expect(x, dc, BV_log=True), steps=50)
        byt = 4 * H * W * (D + 1) * B
        print(json.dumps({"case": "dpv_expect B=%d D=%d %dx%d" % (B, D, H, W), "ms": ms,
                          "algorithmic_GBps": byt / ms / 1e6, "hbm_frac": byt / ms / 1e6 / 8000.0}), flush=True)
    # warp_feature, feedback mode: [B, V=2, 64, 64, 128]
    it = synth.make_batch(4, 4, C=64, D=64, H=64, W=128, V=2, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in it.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ms = timeit(lambda: ops.warp_feature(d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc), steps=50)
    byt = 8 * 64 * 128 * 2 * 64 * 4
    print(json.dumps({"case": "warp_feature B=4 V=2 D=64 64x128", "ms": ms, "algorithmic_GBps": byt / ms / 1e6}), flush=True)


def model_cases():
    """BASELINE config 4 (mono_feedback, 5 chained frames, B=1, 256x512 image -> 64x128 sweep) and the
    plain eval model: whole-model time per frame (MIOpen convs included) next to the hot-path kernels."""
    import time
    from pdepth_amd import harness
    from pdepth_amd.models import get_model
    for nmode in ("default", "default_feedback"):
        model = get_model(synth.default_cfg(nmode), 0)
        synth.seed_weights(model, seed=8)
        model = model.cuda().eval()
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
                          "ms_per_frame": ms, "frames_per_s": 1e3 / ms}), flush=True)


if __name__ == "__main__":
    main()
    model_cases()
