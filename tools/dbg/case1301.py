"""Debug of soak case 1301 (seed 31): cell-list kernels with unsorted random depth candidates, D = 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth
rng = np.random.default_rng(31)
for case in range(1302):
    h_, w_ = int(rng.integers(2, 200)), int(rng.integers(2, 400)); c_, d_, v_ = int(rng.integers(1, 72)), int(rng.integers(1, 161)), int(rng.integers(1, 4))
    b_ = int(rng.integers(1, 3)); pose = ('mono', 'stereo', 'wide', 'identity')[int(rng.integers(0, 4))]
    cxo, cyo = float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2)); k = int(rng.integers(0, 6)); dc = None
    if k == 1: rng.uniform(-0.3, 0.3); rng.uniform(-2.5, 2.5, size=3)
    elif k == 2: rng.uniform(-30, 30, size=3)
    elif k in (3, 4): dc = rng.uniform(0.5, 60.0, size=d_)
H, W, C, D, V, B = h_, w_, c_, d_, v_, b_
print(H, W, C, D, V, B, pose, k)
b = synth.make_batch(5000 + 1301, B, C=C, D=D, H=H, W=W, V=V, pose=pose, cx_off=cxo, cy_off=cyo)
dev = torch.device('cuda')
d = {kk: (v.to(dev) if isinstance(v, torch.Tensor) else v) for kk, v in b.items()}
def cmp(name, dcand, algo='cells'):
    a = ops.sweep_cost(d['ref'], d['src'], d['K'], d['R'], d['t'], d['rays'], d['cxcy'], dcand, 8.0, algo=algo).cpu().numpy()
    fb = pdepth_amd._native.fallback_tiles(B, H, W)
    g = ops.sweep_cost(d['ref'], d['src'], d['K'], d['R'], d['t'], d['rays'], d['cxcy'], dcand, 8.0, algo='direct').cpu().numpy()
    fin = np.isfinite(g); bad = (np.abs(a - g) > 1e-5 * max(1.0, float(np.abs(g[fin]).max()))) & fin
    print(f"{name}: bad {bad.sum()} of {bad.size}, nan mismatch {(np.isnan(a) != np.isnan(g)).sum()}, fallback {fb}")
cmp("unsorted D=128", dc)
cmp("sorted ascending D=128", np.sort(dc))
cmp("sorted descending D=128", np.sort(dc)[::-1].copy())
cmp("unsorted first 64", dc[:64].copy())
cmp("unsorted last 64", dc[64:].copy())
cmp("unsorted D=128, tiled1", dc, 'tiled1')
cmp("standard candidates D=128", b['d_candi'])
cmp("unsorted 100", dc[:100].copy())
