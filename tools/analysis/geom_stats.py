"""Footprint statistics of the plane sweep on the synthetic poses (design aid, numpy, CPU).

For every reference pixel: the sample positions on all D planes (float64 restatement of
homography.py:185-196), the set of source texels its bilinear footprints touch, bounding boxes.
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pdepth_amd
from pdepth_amd import synth

def positions(it, v=0):
    K = it["K"].numpy().astype(np.float64); R = it["R"][v].numpy().astype(np.float64); t = it["t"][v].numpy().astype(np.float64)
    rays = it["rays"].numpy().astype(np.float64)
    d = it["d_candi"]
    H, W = it["ref"].shape[-2:]
    t1 = K @ t; t2 = K @ R @ rays
    P = t1[None, :, None] + t2[None] * d[:, None, None]
    z = P[:, 2] + 1e-10
    u = P[:, 0] / z; vv = P[:, 1] / z
    cx, cy = K[0, 2], K[1, 2]
    ix = ((u - cx) / cx + 1) * W / 2 - 0.5
    iy = ((vv - cy) / cy + 1) * H / 2 - 0.5
    return ix.reshape(-1, H, W), iy.reshape(-1, H, W)

def stats(pose, H=256, W=512, D=64, seed=2000, V=1):
    it = synth.make_item(seed, C=4, D=D, H=H, W=W, V=V, pose=pose)
    for v in range(V):
        ix, iy = positions(it, v)
        x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
        print(f"pose={pose} v={v} t={it['t'][v].numpy()} D={D} {H}x{W}")
        for k0 in (0, 8, 16, 32):
            bx = x0[k0:].max(0) - x0[k0:].min(0) + 2; by = y0[k0:].max(0) - y0[k0:].min(0) + 2
            nb = bx * by
            # distinct cells
            cells = np.zeros((H, W), int)
            key = x0[k0:] * 100000 + y0[k0:]
            ks = np.sort(key, axis=0)
            cells = 1 + (np.diff(ks, axis=0) != 0).sum(0)
            print(f"  planes[{k0},{D}): box w mean {bx.mean():.1f} max {bx.max()}  h mean {by.mean():.1f} max {by.max()}  "
                  f"texels mean {nb.mean():.1f} p50 {np.median(nb):.0f} p90 {np.percentile(nb,90):.0f} max {nb.max()}  "
                  f"frac<=32 {np.mean(nb<=32):.2f} <=48 {np.mean(nb<=48):.2f} <=64 {np.mean(nb<=64):.2f} <=96 {np.mean(nb<=96):.2f}; cells mean {cells.mean():.1f} max {cells.max()}")
        # tile-level: 16x4 tiles, union window of all planes
        for (tw, th) in ((16, 4), (32, 8), (64, 4), (32, 16), (64, 8)):
            xs0 = x0.min(0).reshape(H // th, th, W // tw, tw).min((1, 3)); xs1 = x0.max(0).reshape(H // th, th, W // tw, tw).max((1, 3)) + 1
            ys0 = y0.min(0).reshape(H // th, th, W // tw, tw).min((1, 3)); ys1 = y0.max(0).reshape(H // th, th, W // tw, tw).max((1, 3)) + 1
            win = (xs1 - xs0 + 1) * (ys1 - ys0 + 1)
            print(f"  tile {tw}x{th}: window texels mean {win.mean():.0f} max {win.max()}  amplification {win.mean()/(tw*th):.2f}")

if __name__ == "__main__":
    for pose in ("mono", "stereo"):
        for seed in (2000, 2001):
            stats(pose, seed=seed)
    stats("mono", H=512, W=1024, D=128, seed=5000, V=4)
