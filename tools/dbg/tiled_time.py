"""Times the tiled kernel (algo tiled2 / tiled1) with the library PDEPTH_LIB selects (A/B of tools/variants_tiled.sh builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
def case(name, B, C, D, H, W, V, pose, algo, steps=20):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ms = min(timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo), steps=steps) for _ in range(3))
    print("%-28s %-7s %8.4f ms  fallback %d" % (name, algo, ms, pdepth_amd._native.fallback_tiles(B, H, W)), flush=True)
print(os.environ.get("PDEPTH_LIB", "product library"))
case("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono", "tiled2")
case("cfg3 stereo 256x512", 4, 67, 64, 256, 512, 1, "stereo", "tiled2")
case("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono", "tiled1")
case("cfg2 mono 256x512 auto", 4, 67, 64, 256, 512, 1, "mono", "auto")
