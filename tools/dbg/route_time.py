"""The routed path of the default kernel (NCHW entry; csrc/sweep_dist.hip: "Conditioning") against the gather kernel called
directly, and the cost of the (empty) routed launch on the usual input: python tools/dbg/route_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
out = []
for name, B, D, H, W, V, steps in (("B1_64x128", 1, 64, 64, 128, 1, 100), ("B4_256x512", 4, 64, 256, 512, 1, 20)):
    for off in (0.0, 6.0):
        b = synth.make_batch(2, B, C=67, D=D, H=H, W=W, V=V, pose="mono")
        mu = (torch.rand(67, generator=torch.Generator().manual_seed(5)) * 2 - 1) * off
        b["ref"] += mu[None, :, None, None]; b["src"] += mu[None, None, :, None, None]
        d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        dc = ops.d_candi_tensor(d["d_candi"], "cuda")
        for algo in ("auto", "direct"):
            g = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
            out.append("%s off %.0f %s %.1f us" % (name, off, algo, 1e3 * min(timeit(g, steps=steps) for _ in range(3))))
print("\n".join(out))
