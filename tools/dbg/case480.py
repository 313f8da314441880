"""Debug of soak case 480 (seed 31): which view / planes differ between the tiled kernel and the gather kernel?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth
H, W, C, D, V, B = 195, 286, 22, 83, 3, 1
rng = np.random.default_rng(31)
# replay the draws of tools/soak.py up to case 480
for case in range(481):
    h_, w_ = int(rng.integers(2, 200)), int(rng.integers(2, 400)); c_, d_, v_ = int(rng.integers(1, 72)), int(rng.integers(1, 161)), int(rng.integers(1, 4))
    b_ = int(rng.integers(1, 3)); pose = ('mono', 'stereo', 'wide', 'identity')[int(rng.integers(0, 4))]
    cxo, cyo = float(rng.uniform(-3, 3)), float(rng.uniform(-2, 2)); k = int(rng.integers(0, 6))
    tvec = None
    if k == 1: rng.uniform(-0.3, 0.3); rng.uniform(-2.5, 2.5, size=3)
    elif k == 2: tvec = rng.uniform(-30, 30, size=3)
    elif k in (3, 4): rng.uniform(0.5, 60.0, size=d_)
assert (h_, w_, c_, d_, v_, b_, k) == (H, W, C, D, V, B, 2), (h_, w_, c_, d_, v_, b_, k)
b = synth.make_batch(5000 + 480, B, C=C, D=D, H=H, W=W, V=V, pose=pose, cx_off=cxo, cy_off=cyo)
b['t'][0, 0] = torch.from_numpy(tvec.astype(np.float32))
dev = torch.device('cuda')
d = {kk: (v.to(dev) if isinstance(v, torch.Tensor) else v) for kk, v in b.items()}
def run(algo, views, planes=None):
    src, R, t = d['src'][:, views].contiguous(), d['R'][:, views].contiguous(), d['t'][:, views].contiguous()
    dc = d['d_candi'] if planes is None else d['d_candi'][planes]
    return ops.sweep_cost(d['ref'], src, d['K'], R, t, d['rays'], d['cxcy'], dc, 8.0, algo=algo).cpu().numpy()
def cmp(name, views, planes=None):
    a, g = run('tiled1', views, planes), run('direct', views, planes)
    fin = np.isfinite(g); bad = (np.abs(a - g) > 1e-5 * max(1.0, float(np.abs(g[fin]).max()))) & fin
    kk = np.unique(np.nonzero(bad)[1])
    print(f"{name}: bad {bad.sum()} planes {kk.tolist()[:30]} fallback {pdepth_amd._native.fallback_tiles(B, H, W)}")
cmp("all views", [0, 1, 2])
for v in range(3): cmp(f"view {v} alone", [v])
cmp("views 0,1", [0, 1]); cmp("views 1,2", [1, 2]); cmp("views 0,2", [0, 2])
cmp("all views, planes 0..79", [0, 1, 2], slice(0, 80)); cmp("all views, planes 16..82", [0, 1, 2], slice(16, 83))
cmp("all views, planes 3..82 (D=80)", [0, 1, 2], slice(3, 83))
