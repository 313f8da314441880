"""Where do the soak's worst cases sit against an EXACT evaluation?  For each pinned case (tests/test_soak_regressions.py):
the cost volume / depth of the float32 sample positions (the oracle's, bit-identical in every kernel) evaluated in float64
(bilinear taps, squared differences, sums, softmax, expectation), and the distance of the float32 oracle, of `direct`
(reference op order on the GPU) and of `auto` from it.  Tells rounding noise of the float32 reference (which no
implementation can be asked to reproduce unless it copies the summation order) from error of a kernel.

    python tools/dbg/soak_exact.py            (GPU box)
"""
import os, sys, json
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.join(REPO, "tools"))
import importlib.util
import numpy as np
import torch
import pdepth_amd  # noqa
from pdepth_amd import ops
from oracle import ref_cpu as O
from util import oracle_batch, to_dev
spec = importlib.util.spec_from_file_location("soak_tool", os.path.join(REPO, "tools", "soak.py"))
soak = importlib.util.module_from_spec(spec); spec.loader.exec_module(soak)
sys.path.insert(0, os.path.join(REPO, "tests"))
from test_soak_regressions import CASES, SIGMA


def exact_item(it, sigma):
    """float64 cost [D,h,w], depth [h,w] of one item at the oracle's float32 sample positions"""
    ref, src, K = it["ref"].double(), it["src"].double(), it["K"]
    C, h, w = ref.shape
    V = src.shape[0]
    d32 = np.asarray(it["d_candi"]).astype(np.float32)
    D = len(d32)
    cx, cy = K.numpy()[0, 2], K.numpy()[1, 2]
    cost = torch.zeros(D, h * w, dtype=torch.float64)
    for v in range(V):
        ix, iy = O.sample_coords(K, it["R"][v], it["t"][v], it["rays"], d32, cx, cy, h, w)   # [D, hw] float32
        ix, iy = ix.double(), iy.double()
        x0, y0 = torch.floor(ix), torch.floor(iy)
        fx, fy = ix - x0, iy - y0
        val = torch.zeros(D, C, h * w, dtype=torch.float64)
        sv = src[v].reshape(C, h * w)
        for dy, dx, wt in ((0, 0, (1 - fx) * (1 - fy)), (0, 1, fx * (1 - fy)), (1, 0, (1 - fx) * fy), (1, 1, fx * fy)):
            xx, yy = x0 + dx, y0 + dy
            ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1) & torch.isfinite(ix) & torch.isfinite(iy)
            idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).long()
            idx = torch.where(ok, idx, torch.zeros_like(idx))
            tap = sv[:, idx.reshape(-1)].reshape(C, D, h * w).permute(1, 0, 2)
            val += tap * (wt * ok)[:, None, :]
        nan = ~(torch.isfinite(ix) & torch.isfinite(iy))
        dist = ((val - ref.reshape(1, C, h * w)) ** 2).sum(1)
        dist[nan] = float("nan")
        cost += dist / sigma
    logp = torch.log_softmax(cost, 0)
    depth = (torch.from_numpy(d32.astype(np.float64))[:, None] * logp.exp()).sum(0)
    return cost.reshape(D, h, w), depth.reshape(h, w)


def main():
    dev = torch.device("cuda:0")
    rows = {}
    for log, seed, spc, case, _ in CASES:
        shape, b = soak.replay_case(seed, case, spec=spc, offset=True)
        if shape["H"] * shape["W"] * shape["D"] * shape["C"] > 3.0e8:
            print(log, case, "skipped (size)")
            continue
        ocost, _, odepth = oracle_batch(b, sigma=SIGMA)
        it = {k: (v[0] if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        xcost, xdepth = exact_item(it, SIGMA)
        d = to_dev(b, dev)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], SIGMA)
        fin = torch.isfinite(xdepth)
        row = {"shape": shape, "cost_max": float(xcost[torch.isfinite(xcost)].abs().max())}
        def dist(name, cost, depth):
            ce = (cost.double() - xcost)[torch.isfinite(xcost)].abs()
            de = (depth.double() - xdepth)[fin].abs()
            row[name] = {"cost_max": float(ce.max()), "cost_rms": float((ce ** 2).mean().sqrt()), "depth_max": float(de.max()),
                         "depth_over_1e4": int((de > 1e-4).sum())}
        dist("oracle32", ocost[0], odepth[0])
        for algo in ("direct", "auto", "tiled1"):
            c, _, dp = ops.sweep_dpv(*args, feat_dist="L2", algo=algo, want_cost=True)
            dist(algo, c[0].cpu(), dp[0].cpu())
        # conditioning: depth change per unit of cost error, worst pixel: max_k p_k |d_k - E|
        p = torch.log_softmax(xcost.reshape(xcost.shape[0], -1), 0).exp()
        dk = torch.from_numpy(np.asarray(b["d_candi"]).astype(np.float64))[:, None]
        sens = (p * (dk - xdepth.reshape(1, -1)).abs()).sum(0)
        row["sensitivity_m_per_unit_cost_max"] = float(sens[fin.reshape(-1)].max())
        rows[f"{log}:{case}"] = row
        print(log, case, json.dumps(row))
    json.dump(rows, open(os.path.join(REPO, "gpurun_out", "soak_exact.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
