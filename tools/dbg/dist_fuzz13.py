"""Replays cases of tests/test_hip_parity.py::test_fuzz_tiled_against_gather for algo dist: error map of the failing ones."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
from util import to_dev
dev = torch.device("cuda:0")
rng = np.random.default_rng(2024)
for case in range(120):
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
    args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 7.5)
    cd, ld, dd = ops.sweep_dpv(*args, want_cost=True, algo="direct")
    ca, la, da = ops.sweep_dpv(*args, want_cost=True, algo="dist")
    torch.cuda.synchronize()
    nd = _native._queue_slot(1, H, W, 59)
    e = (ca - cd).abs()
    nanm = torch.isnan(ca) != torch.isnan(cd)
    scale = max(1.0, float(torch.nan_to_num(cd.abs()).max()))
    err = float(torch.nan_to_num(e).max()) / scale
    if err > 2e-5 or nanm.any():
        bad = (torch.nan_to_num(e) / scale > 2e-5) | nanm
        idx = bad.nonzero()
        print("case %d kind %d %dx%d C=%d D=%d V=%d: err %.3e nan-mismatch %d bad %d direct-blocks %d" % (case, kind, H, W, C, D, V, err, int(nanm.sum()), int(bad.sum()), nd))
        print("   planes:", sorted(set(idx[:, 1].tolist()))[:40])
        print("   y:", int(idx[:, 2].min()), int(idx[:, 2].max()), "x:", int(idx[:, 3].min()), int(idx[:, 3].max()))
        for j in idx[:4].tolist(): print("   ", j, "direct", float(cd[tuple(j)]), "dist", float(ca[tuple(j)]))
print("done")
