"""Texel slots per parity role of the fast cell-list kernel (in-lane dedupe, 4 contiguous plane quarters)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from geom_stats import positions
import pdepth_amd
from pdepth_amd import synth
def stats(pose, H=256, W=512, D=64, seed=2000, k0=0):
    it = synth.make_item(seed, C=4, D=max(D, k0 + 64), H=H, W=W, V=1, pose=pose)
    ix, iy = positions(it, 0)
    x0 = np.floor(ix).astype(int)[k0:k0 + 64]; y0 = np.floor(iy).astype(int)[k0:k0 + 64]
    valid = (x0 >= -1) & (x0 <= W - 1) & (y0 >= -1) & (y0 <= H - 1)
    ex = (x0 + 1) & ~1; ox = x0 | 1; ey = (y0 + 1) & ~1; oy = y0 | 1
    tot = np.zeros((4, H, W), int)
    pk = None
    for k in range(64):
        if k % 16 == 0 and k > 0:
            # lane boundary: compare with the previous lane's LAST plane (may be invalid)
            pass
        if k == 0:
            pv = np.zeros((H, W), bool); pex = pox = pey = poy = np.full((H, W), -10**6); pkx = pky = np.full((H, W), -10**6)
        v = valid[k]
        newc = v & ((x0[k] != pkx) | (y0[k] != pky))
        sx = [ex[k] == pex, ox[k] == pox]; sy = [ey[k] == pey, oy[k] == poy]
        for r in range(4):
            tot[r] += newc & ~(sx[r & 1] & sy[r >> 1])
        pkx = np.where(v, x0[k], pkx); pky = np.where(v, y0[k], pky)
        pex = np.where(v, ex[k], pex); pox = np.where(v, ox[k], pox); pey = np.where(v, ey[k], pey); poy = np.where(v, oy[k], poy)
    mx = tot.max(0)
    tm = mx.reshape(H // 4, 4, W // 16, 16).max((1, 3))
    wm = mx.reshape(H, W // 16, 16).max(2)
    print(f"{pose} {H}x{W} planes [{k0},{k0+64}): texels/role mean {tot.mean():.1f} pixel-max mean {mx.mean():.1f} max {mx.max()}; "
          f"wave-max mean {wm.mean():.1f}; tile-max: >16 {np.mean(tm>16):.3f} >20 {np.mean(tm>20):.3f} >24 {np.mean(tm>24):.3f} >28 {np.mean(tm>28):.3f}")
stats("mono"); stats("mono", seed=2001); stats("stereo"); stats("mono", H=64, W=128)
stats("mono", H=512, W=1024, D=128, seed=5000); stats("mono", H=512, W=1024, D=128, seed=5000, k0=64)
