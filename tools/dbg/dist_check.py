"""Quick check of the distance-form sweep kernel (algo=dist) against the gather kernel (algo='direct') and, on small
shapes, the CPU oracle -- plain and with offset (non-centred) features -- then timings."""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit

def run(name, B, C, D, H, W, V, pose, offset=0.0, oracle=False, algo="dist", **kw):
    b = synth.make_batch(7, B, C=C, D=D, H=H, W=W, V=V, pose=pose, **kw)
    if offset:
        g = torch.Generator().manual_seed(5)
        mu = (torch.rand(C, generator=g) * 2 - 1) * offset
        b["ref"] = b["ref"] + mu[None, :, None, None]
        b["src"] = b["src"] + mu[None, None, :, None, None]
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
    c0, l0, z0 = ops.sweep_dpv(*args, algo="direct", want_cost=True)
    c1, l1, z1 = ops.sweep_dpv(*args, algo=algo, want_cost=True)
    torch.cuda.synchronize()
    fb = pdepth_amd._native.fallback_tiles(B, H, W)
    cm = c0.abs().max().item()
    line = "%-40s cost %.3e (rel max %.2e)  logp %.3e  depth %.3e  fallback %d  nan %d" % (
        name, (c0 - c1).abs().max().item(), (c0 - c1).abs().max().item() / cm, (l0 - l1).abs().max().item(),
        (z0 - z1).abs().max().item(), fb, int(torch.isnan(c1).sum().item()))
    if oracle:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from util import oracle_batch
        co, lo, zo = oracle_batch(b)
        line += " | vs oracle: cost %.3e depth %.3e (gather: %.3e)" % ((co - c1.cpu()).abs().max().item(), (zo - z1.cpu()).abs().max().item(),
                                                                    (zo - z0.cpu()).abs().max().item())
    print(line, flush=True)

def tm(name, B, C, D, H, W, V, pose, algo, steps=20):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    f = lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
    ms = min(timeit(f, steps=steps) for _ in range(3))
    line = "%-28s %-7s %8.4f ms  fallback %d" % (name, algo, ms, pdepth_amd._native.fallback_tiles(B, H, W))
    if algo in ("auto", "corr", "dist"):
        ps = ops.pack_source(d["src"], D, algo)
        g = lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
        line += "   packed entry %8.4f ms" % min(timeit(g, steps=steps) for _ in range(3))
    print(line, flush=True)

if "--time-only" not in sys.argv:
    run("tiny C=7 D=8 16x24 V=2", 1, 7, 8, 16, 24, 2, "mono", cx_off=1.3, cy_off=-0.7, oracle=True)
    run("tiny offset 8", 1, 7, 8, 16, 24, 2, "mono", cx_off=1.3, cy_off=-0.7, offset=8.0, oracle=True)
    run("C=67 D=64 64x128 mono", 2, 67, 64, 64, 128, 1, "mono", oracle=True)
    run("C=67 D=64 64x128 mono offset 8", 2, 67, 64, 64, 128, 1, "mono", offset=8.0, oracle=True)
    run("C=67 D=64 64x128 stereo offset 3", 2, 67, 64, 64, 128, 1, "stereo", offset=3.0, oracle=True)
    run("C=67 D=64 64x128 stereo", 2, 67, 64, 64, 128, 1, "stereo")
    run("C=67 D=64 37x83 ragged V=2", 1, 67, 64, 37, 83, 2, "mono")
    run("C=67 D=50 64x96 peaked", 1, 67, 50, 64, 96, 1, "stereo", peaked=True)
    run("C=67 D=128 64x128 V=3", 1, 67, 128, 64, 128, 3, "mono")
    run("C=67 D=100 40x72 wide", 1, 67, 100, 40, 72, 1, "wide")
    run("C=67 D=100 40x72 wide offset 5", 1, 67, 100, 40, 72, 1, "wide", offset=5.0)
    run("C=64 D=64 64x128", 1, 64, 64, 64, 128, 1, "mono")
    run("C=71 D=64 64x128", 1, 71, 64, 64, 128, 1, "mono")
    run("C=67 D=64 256x512 mono B=2", 2, 67, 64, 256, 512, 1, "mono")
    run("C=67 D=64 256x512 stereo B=2", 2, 67, 64, 256, 512, 1, "stereo")
    run("tiled1 offset 8 (guard)", 2, 67, 64, 64, 128, 1, "mono", offset=8.0, oracle=True, algo="tiled1")
    print("guard after tiled1 offset:", pdepth_amd._native.noncentred_guard(2, 64, 128))
tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono", "dist")
tm("cfg3 stereo 256x512", 4, 67, 64, 256, 512, 1, "stereo", "dist")
tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono", "tiled2")
tm("model-real 64x128 B=4", 4, 67, 64, 64, 128, 1, "mono", "dist", steps=50)
tm("model-real 64x128 B=1", 1, 67, 64, 64, 128, 1, "mono", "dist", steps=50)
tm("cfg5 D=128 512x1024 V=4", 2, 67, 128, 512, 1024, 4, "mono", "dist", steps=5)
tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono", "corr")
tm("cfg3 stereo 256x512", 4, 67, 64, 256, 512, 1, "stereo", "corr")
