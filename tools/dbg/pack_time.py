"""Timing of the pre-pass (statistics + pack) and of both entries on the headline shape: python tools/dbg/pack_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
for name, kw in (("headline B=4 256x512 V=1", dict(B=4, C=67, D=64, H=256, W=512, V=1)), ("model-real B=1 64x128", dict(B=1, C=67, D=64, H=64, W=128, V=1)),
                 ("config 5 B=2 V=4 D=128", dict(B=2, C=67, D=128, H=256, W=512, V=4))):
    B = kw.pop("B")
    b = synth.make_batch(2, B, pose="mono", **kw)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    for algo in ("dist",):
        pk = lambda: ops.pack_source(d["src"], kw["D"], algo)
        ps = pk()
        sw = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
        full = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
        t = [min(timeit(f, steps=20) for _ in range(3)) for f in (pk, sw, full)]
        print("%-28s %-5s pack %.4f ms  packed sweep %.4f ms  NCHW entry %.4f ms" % (name, algo, *t), flush=True)
