"""Round 5 rows that need no GPU: the committed bench line of the round, the float64 yardstick
against the float32 oracle, and the inference-only notice of the PackNet host object."""
import json
import os
import warnings

import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import synth
from util import exact_batch, noise_and_explained, oracle_batch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_carries_the_secondary_figures():
    """SURVEY section 8(d) "Ambiguity resolved": the 64x128 model-real figure (B = 1 and B = 4, both entries) and the peaked
    variant ride in the ONE line next to the headline; the line names the kernel that ran, says what ran in front of the
    timed steps and where its PMC traffic record comes from."""
    line = json.loads(open(os.path.join(REPO, "profiles", "r05_bench_line.json")).read().strip().splitlines()[-1])
    assert line["metric"].startswith("depth-volumes/sec") and line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    assert line["roofline"]["sweep_kernel"] == "dist" and 0.05 < line["roofline"]["frac"] < 1
    for key in ("B1_nchw", "B1_packed", "B4_nchw", "B4_packed"):
        row = line["model_real"][key]
        assert row["us_per_call"] > 0 and row["launches"] in (1, 3) and 0 < row["frac"] < 1 and row["volumes_per_s"] > 0, key
    assert line["model_real"]["B1_packed"]["launches"] == 1 and "64x128" in line["model_real"]["shape"]
    assert line["peaked"]["ms_per_step"] > 0 and "correlated" in line["peaked"]["what"]
    assert line["packed_entry"]["max_abs_depth_diff_vs_headline"] == 0.0 and line["packed_entry"]["kernel_ms"] < line["roofline"]["kernel_ms"]
    assert line["cold_start"]["ms_per_step"] > 0 and "untimed" in line["preheat"] and line["preheat_ms"] > 0
    assert line["roofline"]["traffic"] > line["roofline"]["algorithmic_bytes_per_launch"] and "r05" in line["roofline"]["traffic_source"]
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["max_abs_depth_diff_gpu_vs_port_item0"] <= 1e-4


def test_float64_yardstick_agrees_with_the_oracle():
    """oracle.sweep_dpv_exact64 evaluates the reference's formula in float64 at the reference's float32 sample positions: on
    well-conditioned inputs the float32 oracle sits at rounding distance from it (validates the yardstick the soak
    regressions use), border and out-of-image taps included."""
    for pose, V in (("mono", 1), ("wide", 2)):
        b = synth.make_batch(11, 2, C=19, D=24, H=22, W=37, V=V, pose=pose)
        ocost, _, odepth = oracle_batch(b)
        xcost, xdepth, xkappa = exact_batch(b)
        fin = torch.isfinite(xcost)
        assert torch.equal(torch.isfinite(ocost), fin)
        assert float((ocost.double() - xcost)[fin].abs().max()) < 2e-5
        dfin = torch.isfinite(xdepth)
        assert float((odepth.double() - xdepth)[dfin].abs().max()) < 1e-4
        r = noise_and_explained(ocost, odepth, ocost, odepth, xcost, xkappa)
        assert r["noise_max_ratio"] == 1.0 and r["unexplained_m"] == 0.0 and r["kappa_max"] > 0


def test_packnet_host_object_says_that_it_is_inference_only():
    from pdepth_amd.models import get_model
    m = get_model(synth.default_cfg(model_name="packnet"), 0)
    m.attach_networks(torch.nn.Conv2d(3, 4, 1), torch.nn.Conv2d(4, 4, 1))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m._inference_only()
        m._inference_only()
    assert len([x for x in w if "no backward" in str(x.message)]) == 1
    with torch.no_grad(), warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        m._warned_no_grad = False
        m._inference_only()
    assert not w2
