"""Fuzz case 13 of test_fuzz_tiled_against_gather, per view and all views, dist vs direct."""
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
print("case", case, H, W, C, D, V)
def cmp(name, src, R, t):
    args = (d["ref"], src, d["K"], R, t, d["rays"], d["cxcy"], d["d_candi"], 7.5)
    cd, _, _ = ops.sweep_dpv(*args, want_cost=True, algo="direct")
    ca, _, _ = ops.sweep_dpv(*args, want_cost=True, algo="dist")
    torch.cuda.synchronize()
    nd = _native._queue_slot(1, H, W, 59)
    e = torch.nan_to_num((ca - cd).abs())
    bad = e > 1e-3
    print("%-12s err %.3e bad %d direct-blocks %d" % (name, float(e.max()), int(bad.sum()), nd), "rows", sorted(set(bad.nonzero()[:, 2].tolist()))[:30] if bad.any() else "")
cmp("all", d["src"], d["R"], d["t"])
for v in range(V):
    cmp("view %d" % v, d["src"][:, v:v+1].contiguous(), d["R"][:, v:v+1].contiguous(), d["t"][:, v:v+1].contiguous())
for vs in ((0, 1), (1, 2), (0, 2), (1, 0), (2, 1)):
    if max(vs) < V:
        idx = list(vs)
        cmp("views %s" % (vs,), d["src"][:, idx].contiguous(), d["R"][:, idx].contiguous(), d["t"][:, idx].contiguous())
print("---- decomposition of the (0,1) result")
def run(algo, idx):
    args = (d["ref"], d["src"][:, idx].contiguous(), d["K"], d["R"][:, idx].contiguous(), d["t"][:, idx].contiguous(), d["rays"], d["cxcy"], d["d_candi"], 7.5)
    c, _, _ = ops.sweep_dpv(*args, want_cost=True, algo=algo)
    torch.cuda.synchronize()
    return c
c01 = run("dist", [0, 1]); c0 = run("dist", [0]); c1 = run("dist", [1]); g01 = run("direct", [0, 1])
e = (c01 - g01).abs()
bad = (e > 1e-3).nonzero()
print("bad", len(bad))
for j in bad[:: max(1, len(bad) // 12)][:12].tolist():
    t = tuple(j)
    print(j, "dist01 %.5f  gather01 %.5f  dist0 %.5f  dist1 %.5f  | dist01 - dist1 = %.5f" % (float(c01[t]), float(g01[t]), float(c0[t]), float(c1[t]), float(c01[t] - c1[t])))
