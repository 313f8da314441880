"""mfma vs auto (tiled) over a matrix of shapes and poses: which one should ALGO_AUTO pick?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
def tm(B, C, D, H, W, V, pose, algo, steps=10):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    return min(timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo), steps=steps) for _ in range(2))
for (B, D, H, W, V) in ((4, 64, 256, 512, 1), (4, 64, 256, 512, 2), (2, 64, 256, 512, 4), (4, 128, 256, 512, 1), (2, 128, 256, 512, 2), (4, 32, 256, 512, 1),
                        (4, 64, 128, 256, 1), (8, 64, 64, 128, 1), (1, 64, 64, 128, 1), (1, 64, 64, 96, 1), (2, 64, 512, 1024, 1), (2, 128, 512, 1024, 4)):
    for pose in ("mono", "stereo"):
        t = {a: tm(B, 67, D, H, W, V, pose, a) for a in ("mfma", "auto")}
        print("B=%d D=%3d %4dx%-4d V=%d %-6s  mfma %8.4f  auto %8.4f  ratio %.2f" % (B, D, H, W, V, pose, t["mfma"], t["auto"], t["mfma"] / t["auto"]), flush=True)
