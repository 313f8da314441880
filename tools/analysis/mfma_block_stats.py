"""Design aid of sweep_mfma.hip (numpy, CPU): blocks of 16 texels that cover the texels the 64 planes of 16 neighbouring pixels touch, per pixel-block shape."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from geom_stats import positions
import pdepth_amd
from pdepth_amd import synth

def run(pose, H=256, W=512, D=64, seed=2000, V=1, k0=0, k1=None):
    it = synth.make_item(seed, C=4, D=D, H=H, W=W, V=V, pose=pose)
    k1 = k1 or D
    ix, iy = positions(it, 0)
    x0 = np.floor(ix[k0:k1]).astype(int); y0 = np.floor(iy[k0:k1]).astype(int)
    # clamp to image-ish to avoid huge
    x0 = np.clip(x0, -2, W); y0 = np.clip(y0, -2, H)
    res = {}
    for (pw, ph) in ((16,1),(8,2),(4,4)):
        nrun = []; np82 = []; np44 = []; ntex = []; np161=[]
        for by in range(0, H, ph*8):      # subsample blocks for speed
            for bx in range(0, W, pw):
                xs = x0[:, by:by+ph, bx:bx+pw].ravel(); ys = y0[:, by:by+ph, bx:bx+pw].ravel()
                tx = np.concatenate([xs, xs+1, xs, xs+1]); ty = np.concatenate([ys, ys, ys+1, ys+1])
                key = np.unique(ty * 4096 + (tx + 8))
                ntex.append(len(key))
                yy = key // 4096; xx = key % 4096
                # row-run cover
                c = 0
                for r in np.unique(yy):
                    xr = xx[yy == r]
                    c += -(-(xr.max() - xr.min() + 1) // 16)
                nrun.append(c)
                np161.append(len(np.unique(yy * 4096 + xx // 16)))
                np82.append(len(np.unique((yy // 2) * 4096 + xx // 8)))
                np44.append(len(np.unique((yy // 4) * 4096 + xx // 4)))
        f = lambda a: (np.mean(a), np.percentile(a, 90), np.max(a))
        print(f"{pose} planes[{k0},{k1}) pixblock {pw}x{ph}: texels mean %.0f p90 %.0f max %d | runblocks %.1f/%.0f/%d | aligned16x1 %.1f/%.0f/%d | 8x2 %.1f/%.0f/%d | 4x4 %.1f/%.0f/%d" % (*f(ntex), *f(nrun), *f(np161), *f(np82), *f(np44)))

for pose in ("mono", "stereo"):
    run(pose)
run("mono", k0=0, k1=16); run("mono", k0=16, k1=64)
