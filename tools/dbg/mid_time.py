"""Packed-entry time on shapes between the model-real and the headline size (which side of the persistent-queue / workgroup-per-item choice)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
out = []
for (B, H, W) in [tuple(int(x) for x in a.split('x')) for a in sys.argv[1:]] or ((1, 64, 128), (4, 64, 128), (8, 64, 128), (1, 128, 256), (4, 128, 256), (1, 256, 512), (2, 256, 512), (4, 256, 512)):
    b = synth.make_batch(2, B, C=67, D=64, H=H, W=W, V=1, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ps = ops.pack_source(d["src"], 64)
    f = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
    out.append("B=%d %dx%d %.1f" % (B, H, W, min(timeit(f, steps=30) for _ in range(3)) * 1e3))
print(" | ".join(out))
