"""Timings of the correlation-form sweep kernel with the library PDEPTH_LIB selects (A/B of tools/variants_corr.sh builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
def tm(name, B, C, D, H, W, V, pose, algo="corr", steps=20):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    f = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
    ms = min(timeit(f, steps=steps) for _ in range(3))
    line = "%-28s %-7s %8.4f ms  fallback %d" % (name, algo, ms, pdepth_amd._native.fallback_tiles(B, H, W))
    ps = ops.pack_source(d["src"], D, algo)
    g = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
    line += "   packed entry %8.4f ms" % min(timeit(g, steps=steps) for _ in range(3))
    print(line, flush=True)
print(os.environ.get("PDEPTH_LIB", "product library"))
tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono")
tm("cfg3 stereo 256x512", 4, 67, 64, 256, 512, 1, "stereo")
if "--more" in sys.argv:
    tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono", "tiled2")
    tm("model-real 64x128 B=4", 4, 67, 64, 64, 128, 1, "mono", steps=50)
    tm("model-real 64x128 B=1", 1, 67, 64, 64, 128, 1, "mono", steps=50)
    tm("cfg5 D=128 512x1024 V=4", 2, 67, 128, 512, 1024, 4, "mono", steps=5)
