"""NCHW entry of the default kernel (pre-pass + sweep), us per call from HIP events: python nchw_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
out = []
for name, B, D, H, W, V, steps in (("B1_64x128", 1, 64, 64, 128, 1, 100), ("B4_64x128", 4, 64, 64, 128, 1, 100), ("B1_256x512", 1, 64, 256, 512, 1, 50),
                                   ("B4_256x512", 4, 64, 256, 512, 1, 30), ("cfg5", 2, 128, 512, 1024, 4, 5)):
    b = synth.make_batch(2, B, C=67, D=D, H=H, W=W, V=V, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    g = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
    out.append("%s %.1f" % (name, 1e3 * min(timeit(g, steps=steps) for _ in range(4))))
print("  ".join(out))
