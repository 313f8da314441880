"""N back-to-back fused sweep calls of one small shape (for tools/prof_small.sh): python one_sweep.py B H W algo N"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
B, H, W, algo, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
b = synth.make_batch(2, B, C=67, D=64, H=H, W=W, V=1, pose="mono")
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
if algo == "packed":   # the packed-source entry (what the host model runs: the encoder epilogue writes the staging layout)
    ps = ops.pack_source(d["src"], 64)
    f = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
else:
    f = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
for _ in range(10): f()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N): f()
torch.cuda.synchronize()
print("wall per call %.1f us (back-to-back calls, host included)" % ((time.perf_counter() - t0) / N * 1e6))
