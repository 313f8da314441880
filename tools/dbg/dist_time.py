"""Timing only of algo='dist' (packed entry) on the headline shapes: python dist_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
out = []
for pose in ("mono", "stereo"):
    b = synth.make_batch(2, 4, C=67, D=64, H=256, W=512, V=1, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ps = ops.pack_source(d["src"], 64, "dist")
    g = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="dist")
    out.append("%s %.4f" % (pose, min(timeit(g, steps=20) for _ in range(3))))
print("  ".join(out))
