"""Sheared band boxes (sweep_tiled.hip, PDEPTH_SHEAR): slots NC x NR per 32x4 tile against the bounding rectangle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geom_stats import positions
import pdepth_amd
from pdepth_amd import synth

def boxes(ix, iy, k0, k1):
    xl, yl, xh, yh = ix[k1 - 1], iy[k1 - 1], ix[k0], iy[k0]
    x0 = np.floor(np.minimum(xl, xh) - 1e-3); x1 = np.floor(np.maximum(xl, xh) + 1e-3) + 1
    y0 = np.floor(np.minimum(yl, yh) - 1e-3); y1 = np.floor(np.maximum(yl, yh) + 1e-3) + 1
    rect_nc, nr = x1 - x0 + 1, y1 - y0 + 1
    dxs, dys = xh - xl, yh - yl
    horiz = ~(np.abs(dys) > 1e-4)
    inv = np.where(horiz, 0.0, 1.0 / np.where(horiz, 1.0, dys))
    ncs = np.zeros_like(x0); shmax = np.zeros_like(x0)
    for r in range(16):
        rm = y0 + r - 1
        ta, tb = (rm - yl) * inv, (rm + 2 - yl) * inv
        lo = np.where(horiz, 0.0, np.clip(np.minimum(ta, tb), 0, 1)); hi = np.where(horiz, 1.0, np.clip(np.maximum(ta, tb), 0, 1))
        xa, xb = xl + lo * dxs, xl + hi * dxs
        cmin = np.maximum(np.floor(np.minimum(xa, xb) - 1e-3), x0); cmax = np.minimum(np.floor(np.maximum(xa, xb) + 1e-3) + 1, x1)
        rowv = (y0 + r) <= y1
        ncs = np.maximum(ncs, np.where(rowv, cmax - cmin + 1, 0)); shmax = np.maximum(shmax, np.where(rowv, cmin - x0, 0))
    return rect_nc, ncs, nr, shmax

def stats(pose, H=256, W=512, D=64, seed=2000, tw=32, th=4, B=4):
    print(f"== {pose} {H}x{W} D={D} tile {tw}x{th}")
    acc = {}
    for b in range(B):
        it = synth.make_item(seed + b, C=4, D=D, H=H, W=W, V=1, pose=pose)
        ix, iy = positions(it, 0)
        for (k0, k1) in ((16, 64), (8, 64), (0, 64), (32, 64)):
            rc, sc, nr, shm = boxes(ix, iy, k0, k1)
            r = lambda a: a.reshape(H // th, th, W // tw, tw).max(axis=(1, 3))
            NR = r(nr); acc.setdefault((k0, k1), []).append(((r(rc) * NR).ravel(), (r(sc) * NR).ravel(), NR.ravel(), r(shm).ravel()))
    for k, v in acc.items():
        R = np.concatenate([a for a, _, _, _ in v]); S = np.concatenate([a for _, a, _, _ in v]); N = np.concatenate([a for _, _, a, _ in v]); M = np.concatenate([a for _, _, _, a in v])
        print(f"  planes [{k[0]:2d},{k[1]:2d}): rect NX mean {R.mean():6.1f} <=48 {np.mean(R <= 48):.2f} | sheared NX mean {S.mean():6.1f} <=48 {np.mean(S <= 48):.2f} <=32 {np.mean(S <= 32):.2f} <=24 {np.mean(S <= 24):.2f} | NR mean {N.mean():.1f} max {N.max():.0f} | max shift {M.max():.0f}")

for pose in ("mono", "stereo"):
    stats(pose)
