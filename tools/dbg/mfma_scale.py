"""Fixed against per-tile cost of the matrix-pipe sweep (library from PDEPTH_LIB): time of algo='mfma' at B = 1, 2, 4, 8."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
pose = sys.argv[1] if len(sys.argv) > 1 else "mono"
print(os.environ.get("PDEPTH_LIB", "product library"), pose)
for B in (1, 2, 4, 8):
    b = synth.make_batch(2, B, C=67, D=64, H=256, W=512, V=1, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ps = ops.pack_source(d["src"], 64)
    ms = min(timeit(lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="mfma"), steps=20) for _ in range(3))
    print("B=%d  %.4f ms (packed entry: clear + sweep + gather launch)" % (B, ms), flush=True)
