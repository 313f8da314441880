"""Design aid of sweep_mfma.hip (numpy, CPU): 16x1 against 8x2 pixel sub-blocks per 16x4 tile, and how well a one-number rule picks the better one."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from geom_stats import positions
import pdepth_amd
from pdepth_amd import synth

def nblocks(x0, y0, valid):
    xs = x0[valid]; ys = y0[valid]
    if xs.size == 0: return 0
    c = 0
    cells = {}
    for r in np.unique(ys):
        xr = xs[ys == r]; cells[r] = (xr.min(), xr.max())
    rows = set(cells) | set(r + 1 for r in cells)
    for r in rows:
        lo = min(cells[q][0] for q in (r, r - 1) if q in cells); hi = max(cells[q][1] for q in (r, r - 1) if q in cells) + 1
        c += -(-(hi - lo + 1) // 16)
    return c

def run(pose, H=256, W=512, D=64, seed=2000, step=3):
    it = synth.make_item(seed, C=4, D=D, H=H, W=W, V=1, pose=pose)
    ix, iy = positions(it, 0)
    x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
    valid = (x0 >= -1) & (x0 <= W - 1) & (y0 >= -1) & (y0 <= H - 1)
    res = []
    for ty in range(0, H // 4, step):
        for tx in range(0, W // 16, 1):
            a = []
            for s in range(4):   # 16x1 rows
                sl = (slice(None), slice(ty * 4 + s, ty * 4 + s + 1), slice(tx * 16, tx * 16 + 16))
                a.append(nblocks(x0[sl], y0[sl], valid[sl]))
            b = []
            for s in range(4):   # 8x2 sub-blocks
                sy, sx = s // 2, s % 2
                sl = (slice(None), slice(ty * 4 + 2 * sy, ty * 4 + 2 * sy + 2), slice(tx * 16 + 8 * sx, tx * 16 + 8 * sx + 8))
                b.append(nblocks(x0[sl], y0[sl], valid[sl]))
            # predictor from centre pixel near/far displacement
            cy, cxp = ty * 4 + 2, tx * 16 + 8
            dx = abs(ix[0, cy, cxp] - ix[-1, cy, cxp]); dy = abs(iy[0, cy, cxp] - iy[-1, cy, cxp])
            res.append((sum(a), max(a), sum(b), max(b), dx, dy))
    r = np.array(res)
    print(f"{pose}: 16x1 mean blocks/sub {r[:,0].mean()/4:.1f} max {r[:,1].max():.0f} frac(max>14) {np.mean(r[:,1]>14):.2f} >16 {np.mean(r[:,1]>16):.2f} | 8x2 mean {r[:,2].mean()/4:.1f} max {r[:,3].max():.0f} frac>14 {np.mean(r[:,3]>14):.2f} >16 {np.mean(r[:,3]>16):.2f}")
    best = np.minimum(r[:,0], r[:,2]); print(f"   oracle choice mean {best.mean()/4:.2f}")
    for thr in (0.25, 0.5, 1.0, 2.0):
        pick = np.where(r[:,5] * 1.0 > thr * 1.0 + 0 * r[:,4], r[:,2], r[:,0])   # dy > thr rows -> 8x2
        pm = np.where(r[:,5] > thr, r[:,3], r[:,1])
        print(f"   rule dy>{thr}: mean {pick.mean()/4:.2f}  frac(max>14) {np.mean(pm>14):.3f} >16 {np.mean(pm>16):.3f}")
    for f in (0.1, 0.2, 0.4):
        sel = r[:,5] > f * r[:,4] + 0.5
        pick = np.where(sel, r[:,2], r[:,0]); pm = np.where(sel, r[:,3], r[:,1])
        print(f"   rule dy>{f}*dx+0.5: mean {pick.mean()/4:.2f}  frac(max>14) {np.mean(pm>14):.3f} >16 {np.mean(pm>16):.3f}")

run("mono"); run("stereo"); run("mono", seed=2002)
