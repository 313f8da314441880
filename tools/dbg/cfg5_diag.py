"""config 5 (B=2, V=4, D=128, 512x1024): time per call of the distance-form and the correlation-form kernel (packed entry) and
the number of pixel-block passes the distance-form kernel evaluated directly (more texel blocks than its LDS holds)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
from bench_kernels import timeit
for (H, W) in ((256, 512), (512, 1024)):
    b = synth.make_batch(2, 2, C=67, D=128, H=H, W=W, V=4, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    for algo in ("dist", "corr"):
        ps = ops.pack_source(d["src"], 128, algo)
        f = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
        ms = min(timeit(f, steps=10) for _ in range(2))
        f(); torch.cuda.synchronize()
        print("%dx%d %-5s %.3f ms per call (%.3f per volume)  passes evaluated directly: %d of %d" % (
            H, W, algo, ms, ms / 2, _native.fallback_tiles(2, H, W), 2 * 4 * 2 * (H * W // 16)), flush=True)
