"""Fuzz case 13, views (0,1): which samples differ -- per pixel block / plane pattern."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
from util import to_dev
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)
for case in range(14):
    H, W = int(rng.integers(3, 70)), int(rng.integers(3, 110))
    C, D, V = int(rng.integers(1, 12)), int(rng.integers(1, 80)), int(rng.integers(1, 4))
    b = synth.make_batch(60 + case, 1, C=C, D=D, H=H, W=W, V=V, pose="mono", cx_off=float(rng.uniform(-2, 2)), cy_off=float(rng.uniform(-1, 1)))
    kind = case % 4
    if kind == 1:
        ang = rng.uniform(-0.25, 0.25, size=3)
        cz, sz = np.cos(ang[2]), np.sin(ang[2])
        Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float32)
        b["R"][0, 0] = torch.from_numpy(Rz) @ b["R"][0, 0]
        b["t"][0, 0] = torch.from_numpy(rng.uniform(-2.5, 2.5, size=3).astype(np.float32))
    elif kind == 2:
        b["t"][0, 0] = torch.from_numpy(rng.uniform(-30, 30, size=3).astype(np.float32))
    elif kind == 3:
        b["d_candi"] = np.sort(rng.uniform(0.5, 60.0, size=D))[::-1].copy() if case % 8 == 3 else rng.uniform(2.0, 50.0, size=D)
d = to_dev(b, dev)
def run(algo, idx):
    args = (d["ref"], d["src"][:, idx].contiguous(), d["K"], d["R"][:, idx].contiguous(), d["t"][:, idx].contiguous(), d["rays"], d["cxcy"], d["d_candi"], 7.5)
    c, _, _ = ops.sweep_dpv(*args, want_cost=True, algo=algo)
    torch.cuda.synchronize()
    return c
c01 = run("dist", [0, 1]); g01 = run("direct", [0, 1]); c0 = run("dist", [0]); g1 = run("direct", [1])
bad = ((c01 - g01).abs() > 1e-3)[0]       # [D, H, W]
print("bad planes per plane index:", [int(bad[k].sum()) for k in range(D)])
pix = bad.any(dim=0)
print("bad pixel map (rows 0..23, '#' = bad):")
for y in range(24):
    print("".join("#" if pix[y, x] else "." for x in range(W)))
# view 0's contribution inside the two-view run
v0 = (c01 - g1)[0]
rel = (v0 - c0[0]).abs()
print("max |(c01 - gather1) - dist0|:", float(rel.max()))
ys, xs = 2, 40
print("pixel (2,40): planes: two-view view-0 part vs alone")
for k in range(D): print(k, "%.5f %.5f" % (float(v0[k, ys, xs]), float(c0[0, k, ys, xs])))
