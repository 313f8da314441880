"""Correctness (vs the gather kernel) + timing of algo='dist' on the bench shapes: python dist_quick.py [algo]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
from bench_kernels import timeit
algo = sys.argv[1] if len(sys.argv) > 1 else "dist"
def check(name, B, C, D, H, W, V, pose, **kw):
    b = synth.make_batch(7, B, C=C, D=D, H=H, W=W, V=V, pose=pose, **kw)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
    c0, l0, z0 = ops.sweep_dpv(*args, algo="direct", want_cost=True)
    c1, l1, z1 = ops.sweep_dpv(*args, algo=algo, want_cost=True)
    torch.cuda.synchronize()
    print("%-34s cost %.3e logp %.3e depth %.3e direct-blocks %d nan %d" % (name, (c0 - c1).abs().max().item(), (l0 - l1).abs().max().item(),
          (z0 - z1).abs().max().item(), _native._queue_slot(B, H, W, 59), int(torch.isnan(c1).sum().item())), flush=True)
def tm(name, B, C, D, H, W, V, pose, steps=20):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    f = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
    ms = min(timeit(f, steps=steps) for _ in range(3))
    ps = ops.pack_source(d["src"], D, algo)
    g = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
    print("%-28s %-6s NCHW %8.4f ms   packed %8.4f ms" % (name, algo, ms, min(timeit(g, steps=steps) for _ in range(3))), flush=True)
check("64x128 mono B=2", 2, 67, 64, 64, 128, 1, "mono")
check("37x83 ragged V=2", 1, 67, 64, 37, 83, 2, "mono")
check("D=128 V=3 64x128", 1, 67, 128, 64, 128, 3, "mono")
check("wide D=100 40x72", 1, 67, 100, 40, 72, 1, "wide")
check("C=7 D=8 16x24 V=2", 1, 7, 8, 16, 24, 2, "mono", cx_off=1.3, cy_off=-0.7)
check("C=33 D=64 64x128", 1, 33, 64, 64, 128, 1, "mono")
check("256x512 mono B=2", 2, 67, 64, 256, 512, 1, "mono")
check("256x512 stereo B=2", 2, 67, 64, 256, 512, 1, "stereo")
tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono")
tm("cfg3 stereo 256x512", 4, 67, 64, 256, 512, 1, "stereo")
tm("model-real 64x128 B=4", 4, 67, 64, 64, 128, 1, "mono", steps=50)
tm("model-real 64x128 B=1", 1, 67, 64, 64, 128, 1, "mono", steps=50)
tm("cfg5 D=128 512x1024 V=4", 2, 67, 128, 512, 1024, 4, "mono", steps=5)
