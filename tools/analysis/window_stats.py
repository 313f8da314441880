"""Window size distribution of the cell-list kernel's 16x4 tiles (pitch = 8 mod 16 rounding included)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geom_stats import positions
import pdepth_amd
from pdepth_amd import synth
def stats(pose, H=256, W=512, D=64, seed=2000, tw=16, th=4):
    it = synth.make_item(seed, C=4, D=D, H=H, W=W, V=1, pose=pose)
    ix, iy = positions(it, 0)
    x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
    valid = (x0 >= -1) & (x0 <= W - 1) & (y0 >= -1) & (y0 <= H - 1)
    big = 10**6
    xa = np.where(valid, x0, big).min(0); xb = np.where(valid, x0, -big).max(0)
    ya = np.where(valid, y0, big).min(0); yb = np.where(valid, y0, -big).max(0)
    r = lambda a, f: f(f(a.reshape(H // th, th, W // tw, tw), 1), 2)
    wx0 = r(xa, np.min); wx1 = r(xb, np.max); wy0 = r(ya, np.min); wy1 = r(yb, np.max)
    pitch = ((wx1 - wx0 + 2 + 7) & ~15) + 8
    wr = wy1 - wy0 + 2
    tex = pitch * wr
    # cells per pixel
    key = np.where(valid, x0 * 100000 + y0, -1)
    cells = (np.diff(key, axis=0) != 0).sum(0) + 1
    cw = r(cells, np.max)
    print(f"{pose} {H}x{W} D={D}: window texels mean {tex.mean():.0f} p50 {np.median(tex):.0f} p90 {np.percentile(tex,90):.0f} max {tex.max()} "
          f"frac>512 {np.mean(tex>512):.2f} >640 {np.mean(tex>640):.2f} >768 {np.mean(tex>768):.2f} >1024 {np.mean(tex>1024):.2f}; "
          f"cells/pixel mean {cells.mean():.1f} tile-max mean {cw.mean():.1f} max {cw.max()} frac tile-max>24 {np.mean(cw>24):.2f} >16 {np.mean(cw>16):.2f}")
for pose in ("mono", "stereo"):
    stats(pose)
stats("mono", seed=2001)
stats("mono", H=64, W=128)
stats("mono", H=512, W=1024, D=128, seed=5000)
