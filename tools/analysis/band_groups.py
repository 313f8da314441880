"""Box sizes of candidate band groups per 16x4 (and 32x4) tile: what would a second / third band group cost?

For plane ranges [k0, k1): per pixel the box spanned by the positions at the two ends (the position is monotone along
the epipolar line), per tile NC x NR = max over its pixels, window = union of the boxes."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geom_stats import positions
import pdepth_amd
from pdepth_amd import synth

def stats(pose, H=256, W=512, D=64, seed=2000, tw=32, th=4, B=4):
    print(f"== {pose} {H}x{W} D={D} tile {tw}x{th}")
    rows = {}
    for b in range(B):
        it = synth.make_item(seed + b, C=4, D=D, H=H, W=W, V=1, pose=pose)
        ix, iy = positions(it, 0)                       # [D, H, W]
        for (k0, k1) in ((16, 64), (12, 64), (8, 64), (4, 64), (0, 64), (24, 64), (32, 64), (0, 16), (0, 8), (8, 16), (0, 4), (4, 8), (8, 12), (12, 16), (16, 32)):
            if k1 > D: continue
            xa = np.floor(np.minimum(ix[k0], ix[k1 - 1]) - 1e-3); xb = np.floor(np.maximum(ix[k0], ix[k1 - 1]) + 1e-3) + 1
            ya = np.floor(np.minimum(iy[k0], iy[k1 - 1]) - 1e-3); yb = np.floor(np.maximum(iy[k0], iy[k1 - 1]) + 1e-3) + 1
            nc = (xb - xa + 1); nr = (yb - ya + 1)
            r = lambda a, f: f(f(a.reshape(H // th, th, W // tw, tw), 1), 2)
            NC = r(nc, np.max); NR = r(nr, np.max); NX = NC * NR
            wc = r(xa, np.max) + NC - r(xa, np.min); wr = r(ya, np.max) + NR - r(ya, np.min)
            rows.setdefault((k0, k1), []).append((NX.ravel(), (wc * wr).ravel()))
    for (k0, k1), v in rows.items():
        NX = np.concatenate([a for a, _ in v]); WT = np.concatenate([w for _, w in v])
        print(f"  planes [{k0:2d},{k1:2d}): box texels mean {NX.mean():5.1f} p50 {np.median(NX):4.0f} p90 {np.percentile(NX, 90):4.0f} max {NX.max():4.0f}"
              f"  frac<=24 {np.mean(NX <= 24):.2f} <=32 {np.mean(NX <= 32):.2f} <=48 {np.mean(NX <= 48):.2f} <=64 {np.mean(NX <= 64):.2f} | window mean {WT.mean():5.0f} p90 {np.percentile(WT, 90):5.0f}")

for pose in ("mono", "stereo"):
    stats(pose)
stats("mono", tw=16)
