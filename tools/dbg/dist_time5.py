"""Timing only of algo='dist' (packed entry): headline shapes + config 5's share + model-real: python dist_time5.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
out = []
for name, B, D, H, W, V, pose, steps in (("mono", 4, 64, 256, 512, 1, "mono", 20), ("stereo", 4, 64, 256, 512, 1, "stereo", 20),
                                         ("cfg5", 2, 128, 512, 1024, 4, "mono", 5), ("b1", 1, 64, 64, 128, 1, "mono", 50),
                                         ("b1_256", 1, 64, 256, 512, 1, "mono", 30)):
    b = synth.make_batch(2, B, C=67, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ps = ops.pack_source(d["src"], D, "dist")
    g = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="dist")
    out.append("%s %.4f (direct %d)" % (name, min(timeit(g, steps=steps) for _ in range(3)), pdepth_amd._native.fallback_tiles(B, H, W)))
print("  ".join(out))
