"""Diagnostic: which rounding order do torch-CPU matmul / grid_sample use on THIS host?"""
import itertools
import numpy as np, torch
import torch.nn.functional as F
f32 = np.float32
def fma(a, b, c): return (np.asarray(a, f32).astype(np.float64) * np.asarray(b, f32).astype(np.float64) + np.asarray(c, f32).astype(np.float64)).astype(f32)
print("torch", torch.__version__, "cpu capability", torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
torch.manual_seed(0)
perms = list(itertools.permutations(range(3)))
def variants(a, b):  # a[3], b[3,N] -> dict name->result[N]
    out = {}
    for p in perms:
        p0, p1, p2 = [(a[i] * b[i]).astype(f32) for i in p]
        out["nofma%s" % (p,)] = ((p0 + p1).astype(f32) + p2).astype(f32)
        out["fma%s" % (p,)] = fma(a[p[2]], b[p[2]], fma(a[p[1]], b[p[1]], p0))
        out["mix%s" % (p,)] = (fma(a[p[1]], b[p[1]], p0) + p2).astype(f32)
    return out
for N in (1, 3, 8, 384, 8192, 131072):
    for nt in (1, torch.get_num_threads()):
        torch.set_num_threads(nt)
        score = {}
        for it in range(6):
            A = torch.randn(3, 3) * 100; B = torch.randn(3, N)
            C = (A.matmul(B) if N > 1 else A.matmul(B[:, 0]).reshape(3, 1)).numpy()
            for i in range(3):
                for k, v in variants(A.numpy()[i], B.numpy()).items():
                    score[k] = score.get(k, 0) + int((v != C[i]).sum())
        best = sorted(score.items(), key=lambda kv: kv[1])[:3]
        print("N=%6d threads=%3d best:" % (N, nt), best)
# grid_sample un-normalise + interpolation
h, w, D = 60, 100, 4
g = (torch.rand(D, h, w, 2) * 2.2 - 1.1)
img = torch.randn(1, 1, h, w)
out = F.grid_sample(img.repeat(D, 1, 1, 1), g, mode="bilinear", padding_mode="zeros", align_corners=False).numpy()[:, 0]
gx, gy = g[..., 0].numpy(), g[..., 1].numpy(); im = img.numpy()[0, 0]
def tap(xi, yi):
    m = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
    return np.where(m, im[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)], f32(0))
for unv, itv in itertools.product("ABC", "NF"):
    if unv == "A": ix = (((gx + f32(1)) * f32(w) - f32(1)) / f32(2)).astype(f32); iy = (((gy + f32(1)) * f32(h) - f32(1)) / f32(2)).astype(f32)
    elif unv == "B": ix = (((gx + f32(1)).astype(f32) * f32(w / 2)).astype(f32) - f32(0.5)).astype(f32); iy = (((gy + f32(1)).astype(f32) * f32(h / 2)).astype(f32) - f32(0.5)).astype(f32)
    else: ix = fma((gx + f32(1)).astype(f32), f32(w / 2), f32(-0.5)); iy = fma((gy + f32(1)).astype(f32), f32(h / 2), f32(-0.5))
    x0 = np.floor(ix); y0 = np.floor(iy)
    wx = (ix - x0).astype(f32); ex = (f32(1) - wx).astype(f32); ny = (iy - y0).astype(f32); sy = (f32(1) - ny).astype(f32)
    nw = (sy * ex).astype(f32); ne = (sy * wx).astype(f32); sw = (ny * ex).astype(f32); se = (ny * wx).astype(f32)
    xi = x0.astype(np.int64); yi = y0.astype(np.int64)
    a, b, c, d = tap(xi, yi), tap(xi + 1, yi), tap(xi, yi + 1), tap(xi + 1, yi + 1)
    if itv == "N": r = ((((a * nw).astype(f32) + (b * ne).astype(f32)).astype(f32) + (c * sw).astype(f32)).astype(f32) + (d * se).astype(f32)).astype(f32)
    else: r = fma(d, se, fma(c, sw, fma(b, ne, (a * nw).astype(f32))))
    print("grid_sample unnorm", unv, "interp", itv, "mismatch", int((r != out).sum()), "of", out.size)
