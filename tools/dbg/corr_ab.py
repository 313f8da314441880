"""Packed-entry and NCHW-entry timings (mono / stereo, config 2 shape) of several library builds in one process each: python corr_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
out = []
for pose in ("mono", "stereo"):
    b = synth.make_batch(2, 4, C=67, D=64, H=256, W=512, V=1, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ps = ops.pack_source(d["src"], 64, "corr")
    g = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="corr")
    f = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="corr")
    out.append("%s packed %.4f nchw %.4f" % (pose, min(timeit(g, steps=20) for _ in range(3)), min(timeit(f, steps=20) for _ in range(3))))
print("%-46s %s" % (os.environ.get("PDEPTH_LIB", "product"), "  ".join(out)), flush=True)
