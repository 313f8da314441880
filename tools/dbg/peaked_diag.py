"""The headline call on N(0,1) and on the peaked (correlated) features: time per call (packed entry) and passes evaluated directly."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
from bench_kernels import timeit
for pose in ("mono", "stereo"):
    for peaked in (False, True):
        b = synth.make_batch(2, 4, C=67, D=64, H=256, W=512, V=1, pose=pose, peaked=peaked)
        d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        dc = ops.d_candi_tensor(d["d_candi"], "cuda")
        ps = ops.pack_source(d["src"], 64)
        f = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
        g = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
        ms = min(timeit(f, steps=20) for _ in range(3)); ms2 = min(timeit(g, steps=20) for _ in range(3))
        f(); torch.cuda.synchronize()
        print("%-6s peaked=%-5s packed %.4f ms  NCHW %.4f ms  direct passes %d  src mean %.3f std %.3f" % (
            pose, peaked, ms, ms2, _native.fallback_tiles(4, 256, 512), float(d["src"].mean()), float(d["src"].std())), flush=True)
