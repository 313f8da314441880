"""GPU suite: the worst cases of the soak runs, pinned against the WHOLE-IMAGE oracle.

tools/soak.py draws random shapes / poses / candidates and compares every kernel with the gather kernel (and the oracle at
160 sampled pixels).  The cases below are the ones the round-4 soaks reported furthest out (gpurun_out/soak{51,52,61,62}.log;
seeds 5150, 5151, 777, 778 -- recovered from the logged shapes --, all with SOAK_OFFSET=1: every one of them is an even case,
i.e. features with per-channel offsets of up to 8 sigma: costs of 200 .. 760).  Here each is evaluated over the whole image by
the CPU oracle (oracle/ref_cpu.py: the reference's op order, warping/homography.py:98-135 + models/packnet.py:380-394 +
utils/img_utils.py:52-61), and by the float64 evaluation of the same formula at the same sample positions
(oracle.sweep_dpv_exact64), which says how far the float32 reference itself is from what it computes.

What round 5 found (profiles/r05_soak_summary.txt):
  * `direct` summed the channels in sequence, ATen sums runs of 16 (cascade_sum): 2.9e-4 m from the oracle at single
    pixels.  With ATen's order (csrc/sweep_direct.hip) it is the oracle's cost bit for bit nearly everywhere and within
    3.5e-5 m on all six cases -- asserted below at the north star's 1e-4 m, unscaled, also where candidates reach 60 m.
  * the float32 oracle is itself up to 3.2e-4 m from the exact value on these inputs (soak51 case 12: 68 pixels beyond
    1e-4 m): the depth is ill-conditioned there, and within 1e-4 m of the reference means rounding like the reference.
    Round 5 held `auto` to "no noisier than the reference + explained" here; since round 6 the default kernel measures the
    conditioning of an item itself and hands such items to the gather kernel (NCHW entry): `auto` is held to the plain 1e-4 m.
The measured numbers are written to gpurun_out/soak_regressions.json (tools/soak_summary.py formats them)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import _native, ops
from util import DEPTH_ATOL, exact_batch, noise_and_explained, oracle_batch, to_dev

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIGMA = 8.0   # (what tools/soak.py runs with)

# (log, seed, SOAK_SPEC, case, what the soak reported)
CASES = [
    ("soak62", 778, True, 98, "corr vs gather 3.8e-4 m"),
    ("soak62", 778, True, 306, "corr vs oracle (160 px) 3.2e-4 m raw, 2.1e-4 scaled"),
    ("soak52", 5151, True, 202, "corr vs gather 3.5e-4 m"),
    ("soak51", 5150, False, 12, "tiled1 vs gather 3.7e-4 m, vs oracle (160 px) 1.3e-4 m"),
    ("soak61", 777, False, 1846, "corr vs gather 5.8e-4 m raw"),
    ("soak61", 777, False, 1848, "tiled1 vs oracle (160 px) 1.6e-4 m raw"),
]


@pytest.fixture(scope="module")
def soak():
    spec = importlib.util.spec_from_file_location("soak_tool", os.path.join(REPO, "tools", "soak.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    return torch.device("cuda:0")


def _record(key, row):
    path = os.path.join(REPO, "gpurun_out", "soak_regressions.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    try:
        data = json.load(open(path))
    except (OSError, ValueError):
        data = {}
    data[key] = row
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)


@pytest.mark.parametrize("log,seed,spec,case,reported", CASES, ids=[f"{c[0]}-{c[3]}" for c in CASES])
def test_soak_worst_case_against_the_whole_image_oracle(dev, soak, log, seed, spec, case, reported):
    shape, b = soak.replay_case(seed, case, spec=spec, offset=True)
    ocost, _, odepth = oracle_batch(b, sigma=SIGMA)
    xcost, xdepth, xkappa = exact_batch(b, sigma=SIGMA)
    d = to_dev(b, dev)
    args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], SIGMA)
    fin = torch.isfinite(odepth)
    oe = (odepth.double() - xdepth)[fin].abs()
    row = dict(shape=shape, reported=reported, candidates_to_m=round(float(np.max(np.abs(b["d_candi"]))), 2), pixels=int(fin.numel()),
               cost_max=float(xcost[torch.isfinite(xcost)].abs().max()),
               oracle_vs_exact=dict(max_m=float(oe.max()), over_1e4=int((oe > DEPTH_ATOL).sum())))
    res = {}
    for algo in ("auto", "tiled1", "direct"):
        cost, _, depth = ops.sweep_dpv(*args, feat_dist="L2", algo=algo, want_cost=True)
        cost, depth = cost.cpu(), depth.cpu()
        if algo == "auto":
            row["auto_blocks_off_the_fast_path"] = _native.fallback_tiles(shape["B"], shape["H"], shape["W"])
        assert torch.equal(torch.isfinite(depth), fin), f"{algo}: finiteness of the depth differs from the oracle"
        e = (depth - odepth)[fin].abs()
        res[algo] = dict(max_m=float(e.max()) if e.numel() else 0.0, p999_m=float(torch.quantile(e.double(), 0.999)) if e.numel() else 0.0,
                         over_1e4=int((e > DEPTH_ATOL).sum()), **noise_and_explained(cost, depth, ocost, odepth, xcost, xkappa))
        row[algo] = res[algo]
    _record(f"{log}:{case}", row)
    tag = f"{log} case {case} {shape}"
    # the gather kernel rounds like the reference: the north star as it stands
    assert res["direct"]["max_m"] <= DEPTH_ATOL, f"{tag}: direct is {res['direct']['max_m']:.3e} m from the whole-image oracle"
    # the default kernel (NCHW entry): these items are ill-conditioned by its own measure (csrc/sweep_dist.hip: "Conditioning") and
    # go to the gather kernel -- the north star as it stands, for every selector
    a = res["auto"]
    assert a["max_m"] <= DEPTH_ATOL, f"{tag}: auto is {a['max_m']:.3e} m from the whole-image oracle"
    # ... and so does the LDS-tiled kernel (what `auto` runs for the L1 metric, C > 72, D > 128), whose pre-pass applies the same measure
    assert res["tiled1"]["max_m"] <= DEPTH_ATOL, f"{tag}: tiled1 is {res['tiled1']['max_m']:.3e} m from the whole-image oracle"
    assert row["auto_blocks_off_the_fast_path"] > 0, f"{tag}: not routed"
