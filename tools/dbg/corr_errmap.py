"""Where does the correlation-form kernel differ from the gather kernel?  (diagnostic: error map statistics)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
B, H, W = 1, int(sys.argv[1]), int(sys.argv[2])
pose = sys.argv[3] if len(sys.argv) > 3 else "mono"
b = synth.make_batch(7, B, C=67, D=64, H=H, W=W, V=1, pose=pose)
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0)
c0, l0, z0 = ops.sweep_dpv(*args, algo="direct", want_cost=True)
c1, l1, z1 = ops.sweep_dpv(*args, algo="corr", want_cost=True)
torch.cuda.synchronize()
e = (c0 - c1).abs()[0]          # [D,H,W]
bad = e > 1e-3
print("bad fraction", bad.float().mean().item(), "fallback", pdepth_amd._native.fallback_tiles(B, H, W))
print("bad per plane (first 16):", [round(x, 3) for x in bad.float().mean(dim=(1, 2))[:16].tolist()])
print("bad per y%4:", [round(x, 3) for x in bad.float().mean(dim=0).view(H // 4, 4, W).mean(dim=(0, 2)).tolist()])
print("bad per x%16:", [round(x, 3) for x in bad.float().mean(dim=0).view(H, W // 16, 16).mean(dim=(0, 1)).tolist()])
pm = bad.float().mean(dim=0)   # per pixel
print("rows with bad pixels:", (pm.mean(dim=1) > 0).sum().item(), "cols:", (pm.mean(dim=0) > 0).sum().item())
ys, xs = torch.nonzero(pm > 0, as_tuple=True)
if len(ys): print("first bad pixels", list(zip(ys[:8].tolist(), xs[:8].tolist())), "err there", e[:, ys[0], xs[0]][:8].tolist(), "ref", c0[0, :8, ys[0], xs[0]].tolist())
# blocks each 8x2 pixel block needs (row runs cut into 16 texels), against the bad map
import numpy as np
ix, iy = ops.sample_coords(d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, H, W)
ix = ix[0, 0].cpu().numpy(); iy = iy[0, 0].cpu().numpy()   # [D,H,W]
x0 = np.floor(ix).astype(int); y0 = np.floor(iy).astype(int)
pmn = pm.cpu().numpy()
stat = {}
for by in range(0, H, 2):
    for bx in range(0, W, 8):
        xs = x0[:, by:by + 2, bx:bx + 8].ravel(); ys = y0[:, by:by + 2, bx:bx + 8].ravel()
        ok = (xs >= -1) & (xs <= W - 1) & (ys >= -1) & (ys <= H - 1)
        xs, ys = xs[ok], ys[ok]
        nbk = 0; rows = 0
        if len(xs):
            ty = np.concatenate([ys, ys + 1]); tx = np.concatenate([xs, xs + 1])
            for r in np.unique(ty):
                xr = tx[ty == r]; nbk += -(-(xr.max() - xr.min() + 1) // 16)
            rows = len(np.unique(ty))
        isbad = pmn[by:by + 2, bx:bx + 8].mean() > 0
        key = (nbk, isbad)
        stat[key] = stat.get(key, 0) + 1
for k in sorted(stat): print("blocks", k[0], "bad" if k[1] else "ok ", stat[k])
