"""Where does algo='dist' differ from the gather kernel?  Error maps per case."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native

def run(name, B, C, D, H, W, V, pose, offset=0.0, **kw):
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
    c1, l1, z1 = ops.sweep_dpv(*args, algo="dist", want_cost=True)
    torch.cuda.synchronize()
    nd = _native._queue_slot(B, H, W, 59)
    if _native._queue_slot(B, H, W, 60):
        print("   DEBUG neg slots:", _native._queue_slot(B, H, W, 60), [ _native._queue_slot(B, H, W, i) for i in range(32, 43)])
    if _native._queue_slot(B, H, W, 62):
        v = [ _native._queue_slot(B, H, W, i) for i in range(44, 48)]
        import struct
        fl = [struct.unpack("f", struct.pack("i", _native._queue_slot(B, H, W, i)))[0] for i in range(32, 43)]
        print("   scalar ix iy %r %r  packed ix iy %r %r  t2 %r %r %r  packed fw fn %r %r scalar fw fn %r %r" % tuple(fl))
        print("   DEBUG cell mismatches:", _native._queue_slot(B, H, W, 62), "scalar cell (y,x)", v[0] >> 16, (v[0] & 0xffff), "packed cell", v[1] >> 16, v[1] & 0xffff, "lane %d j %d wave %d pt %d" % (v[2] & 255, (v[2] >> 8) & 255, (v[2] >> 16) & 255, v[2] >> 24), "d bits %08x" % (v[3] & 0xffffffff))
    if _native._queue_slot(B, H, W, 61):
        print("   DEBUG non-fit passes:", _native._queue_slot(B, H, W, 61), [ _native._queue_slot(B, H, W, i) for i in range(16, 32)])
    isdir = c1 > 500
    c1 = torch.where(isdir, c1 - 1000, c1)
    print("   values from the direct path:", int(isdir.sum()), " of them bad:", int(((c0 - c1).abs() > 1e-3)[isdir].sum()))
    e = (c0 - c1).abs()
    bad = e > 1e-3
    print("%-36s max err %.3e  bad %d of %d  direct blocks %d" % (name, e.max().item(), int(bad.sum()), bad.numel(), nd))
    if bad.any():
        idx = bad.nonzero()
        print("   bad per item:", [int(bad[i].sum()) for i in range(B)])
        print("   bad planes (count per plane):", [int(bad[:, k].sum()) for k in range(D)][:32], "...")
        ys = idx[:, 2]; xs = idx[:, 3]
        print("   y range", int(ys.min()), int(ys.max()), " x range", int(xs.min()), int(xs.max()))
        # pixel-block pattern: count per (y % 4, x % 16)
        pix = bad.any(dim=1)
        print("   bad pixels", int(pix.sum()), "of", pix.numel())
        for i in range(min(B, 2)):
            rows = pix[i].sum(dim=1)
            print("   item", i, "bad pixels per row (first 24 rows):", rows[:24].tolist())
        j = idx[0].tolist()
        print("   first bad", j, "direct", c0[tuple(j)].item(), "dist", c1[tuple(j)].item())
        j = idx[len(idx) // 2].tolist()
        print("   mid bad", j, "direct", c0[tuple(j)].item(), "dist", c1[tuple(j)].item())

run("64x128 mono", 2, 67, 64, 64, 128, 1, "mono")
run("64x128 mono offset 8", 2, 67, 64, 64, 128, 1, "mono", offset=8.0)
run("64x128 mono offset 1", 2, 67, 64, 64, 128, 1, "mono", offset=1.0)
run("64x128 stereo offset 3", 2, 67, 64, 64, 128, 1, "stereo", offset=3.0)
run("D=128 V=3", 1, 67, 128, 64, 128, 3, "mono")
run("D=64 V=3", 1, 67, 64, 64, 128, 3, "mono")
run("D=128 V=1", 1, 67, 128, 64, 128, 1, "mono")
run("256x512 stereo B=1", 1, 67, 64, 256, 512, 1, "stereo")
run("128x256 stereo B=4", 4, 67, 64, 128, 256, 1, "stereo")
