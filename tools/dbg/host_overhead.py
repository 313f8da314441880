"""Where the host time of a small packed sweep call goes (cProfile over 3000 back-to-back calls, B=1 64x128)."""
import os, sys, cProfile, pstats, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import pdepth_amd
from pdepth_amd import ops, synth
b = synth.make_batch(2, 1, C=67, D=64, H=64, W=128, V=1, pose="mono")
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
ps = ops.pack_source(d["src"], 64)
f = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
for _ in range(200): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3000): f()
torch.cuda.synchronize()
print("wall per call %.1f us" % ((time.perf_counter() - t0) / 3000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(3000): f()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
