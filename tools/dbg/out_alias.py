"""Does it matter where the outputs go?  The packed sweep with the output tensors of the previous call released before the
next one (the caching allocator hands the same 134 MB back: the stores of every call land on the same lines) or kept (two
blocks alternate, as in bench.py's timed loop)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
b = synth.make_batch(2, 4, C=67, D=64, H=256, W=512, V=1, pose="mono")
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
ps = ops.pack_source(d["src"], 64)
keep = [None]
def same():
    ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
def alternate():
    keep[0] = ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
ring = [None] * 8
i = [0]
def ring8():
    ring[i[0] % 8] = ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
    i[0] += 1
for name, f in (("same block", same), ("two blocks alternate", alternate), ("eight blocks in turn", ring8), ("same block", same)):
    print("%-24s %.4f ms" % (name, min(timeit(f, steps=20) for _ in range(3))), flush=True)
