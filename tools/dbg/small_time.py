"""Model-real sizes: GPU time of the pre-pass alone (pack_source), the encoder epilogue (pack_views) and both sweep entries."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
for B in (1, 4):
    b = synth.make_batch(2, B, C=67, D=64, H=64, W=128, V=1, pose="mono")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    feat, rgb = torch.randn(2 * B, 64, 64, 128, device="cuda"), torch.rand(2 * B, 3, 256, 512, device="cuda")
    ps = ops.pack_source(d["src"], 64)
    t = [min(timeit(f, steps=50) for _ in range(3)) * 1e3 for f in (
        lambda: ops.pack_source(d["src"], 64), lambda: ops.pack_views(feat, rgb, 2, 64),
        lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0),
        lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0))]
    print("B=%d 64x128: pack_source %.1f us  pack_views %.1f us  packed sweep %.1f us  NCHW entry %.1f us (events around back-to-back calls: host-bound below ~26 us)" % (B, *t), flush=True)
