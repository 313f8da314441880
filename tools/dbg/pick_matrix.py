"""The shape class in which ALGO_AUTO lets the device pick (V = 1, D <= 64, >= 96 k pixels): matrix-pipe kernel against the
tiled kernel (the faster of its two builds) per shape and pose -- does the rule of pick.hpp still choose the faster one?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
def tm(d, dc, algo, steps=10):
    return min(timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo), steps=steps, warm_ms=15.0) for _ in range(2))
for (B, D, H, W) in ((4, 64, 256, 512), (4, 64, 256, 384), (4, 48, 256, 512), (4, 32, 256, 512), (4, 64, 192, 640), (2, 64, 384, 768), (2, 64, 512, 1024), (1, 64, 256, 512), (8, 64, 256, 512)):
    for pose in ("mono", "stereo", "wide"):
        b = synth.make_batch(2, B, C=67, D=D, H=H, W=W, V=1, pose=pose)
        d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        dc = ops.d_candi_tensor(d["d_candi"], "cuda")
        t = {a: tm(d, dc, a) for a in ("mfma", "tiled1", "tiled2" if D <= 64 else "tiled1", "auto")}
        ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="auto")
        torch.cuda.synchronize()
        choice = _native.sweep_choice(B, H, W)
        best_t = min(t["tiled1"], t["tiled2"])
        print("B=%d D=%2d %3dx%-4d %-6s mfma %7.4f  tiled1 %7.4f  tiled2 %7.4f  auto %7.4f (%s)  %s" % (
            B, D, H, W, pose, t["mfma"], t["tiled1"], t["tiled2"], t["auto"], choice,
            "ok" if (choice == "mfma") == (t["mfma"] <= best_t) else "AUTO PICKS THE SLOWER ONE (%.0f %%)" % (100 * abs(t["mfma"] - best_t) / min(t["mfma"], best_t))), flush=True)
