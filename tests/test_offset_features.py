"""GPU suite: features that are NOT zero-mean, and features with trends.

The L2 cost sum_c (sum_t w_t s_t - r)^2 (warping/homography.py:80-82,129) does not change when the same constant is added
to every reference and source value of a channel -- wherever the four taps lie inside the image.  The correlation form
w^T G w - 2 w.X + |r|^2 that the fast kernels evaluate does: its three terms grow with (mean/std)^2 and cancel.  Round 3's
kernels lost the 1e-4 depth bound at mean/std = 3.  ALGO_AUTO now centres the features (csrc/sweep_dist.hip,
csrc/sweep_pack.hip); the LDS-tiled kernel, which is left for the L1 metric and wide features, switches its correlation-form
plane group off when the pre-pass finds offsets larger than the spread.  Everything here is against the CPU oracle, at the
north-star tolerance: depth 1e-4 m, cost 2e-4 abs + 2e-5 rel."""
import numpy as np
import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import _native, ops, synth
from util import DEPTH_ATOL, oracle_batch, to_dev

pytestmark = pytest.mark.gpu
COST_ATOL, COST_RTOL = 2e-4, 2e-5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    return torch.device("cuda:0")


def _offset_batch(kind, pose, B=2, C=67, D=64, H=64, W=128, V=1, seed=9):
    """N(0,1) features of synth.make_batch plus a per-channel offset: 'uniform8' mu_c ~ U(-8, 8); 'relu' = max(x + 1.5, 0)
    (all positive, like the output of a ReLU); 'ramp' adds a vertical ramp on top of the offset (the sampled rows of the
    statistics kernel see the mean, not the trend: what is left of it after centring is spread, and the rounding error of the
    correlation form grows with the spread -- a ramp of +-2 on unit-variance features costs the last 3e-5 of the budget,
    DESIGN.md section 4 states the domain)."""
    b = synth.make_batch(seed, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    g = torch.Generator().manual_seed(1234 + seed)
    if kind == "uniform8":
        mu = (torch.rand(C, generator=g) * 2 - 1) * 8.0
        b["ref"] = b["ref"] + mu[None, :, None, None]
        b["src"] = b["src"] + mu[None, None, :, None, None]
    elif kind == "relu":
        b["ref"] = torch.clamp(b["ref"] + 1.5, min=0.0)
        b["src"] = torch.clamp(b["src"] + 1.5, min=0.0)
    elif kind.startswith("ramp"):   # 'ramp' = +-1 sigma, 'ramp2' / 'ramp4' = +-2 / +-4 sigma
        mu = (torch.rand(C, generator=g) * 2 - 1) * 4.0
        ramp = torch.linspace(-1.0, 1.0, H)[None, None, :, None] * float(kind[4:] or 1)
        b["ref"] = b["ref"] + mu[None, :, None, None] + ramp
        b["src"] = b["src"] + mu[None, None, :, None, None] + ramp[:, None]
    else:
        raise ValueError(kind)
    return b


def _check(b, dev, algo, **kw):
    ocost, ologp, odepth = oracle_batch(b)
    d = to_dev(b, dev)
    cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0,
                                      algo=algo, want_cost=True, **kw)
    np.testing.assert_allclose(cost.cpu().numpy(), ocost.numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=algo)
    err = (depth.cpu() - odepth).abs().max().item()
    assert err <= DEPTH_ATOL, f"{algo}: depth differs from the oracle by {err:.3e}"
    return err


@pytest.mark.parametrize("pose", ["mono", "stereo"])
@pytest.mark.parametrize("kind", ["uniform8", "relu", "ramp"])
def test_offset_features_against_the_oracle(dev, kind, pose):
    """64x128, C=67, D=64: every implementation ALGO_AUTO can run, on features with per-channel means of up to 8 standard
    deviations."""
    b = _offset_batch(kind, pose)
    for algo in ("auto", "dist", "tiled1", "tiled2", "direct"):
        _check(b, dev, algo)
        # the tiled kernel only gets there because the pre-pass switches its correlation-form plane group off
        if kind == "uniform8" and algo.startswith("tiled"):
            assert _native.noncentred_guard(2, 64, 128) is True


@pytest.mark.parametrize("pose", ["mono", "stereo"])
@pytest.mark.parametrize("kind", ["ramp2", "ramp4"])
def test_trends_are_evaluated_in_the_reference_form(dev, kind, pose):
    """A vertical ramp of +-2 and +-4 standard deviations under the offsets: what a constant per channel cannot remove.  The
    rounding error of every fast form grows with the energy that is left (round 4's default lost the 1e-4 m at +-2); the
    reference's own form (warping/homography.py:80-82,129) does not.  ALGO_AUTO decides per batch item ON THE DEVICE from the
    statistics of the pre-pass (csrc/sweep_dist.hip, the guard): such items take the direct evaluation inside the same launch --
    the diagnostics counter says so -- and meet the north-star bound like `direct` itself; an item of plain N(0,1) features in
    the same batch keeps the fast path."""
    b = _offset_batch(kind, pose)
    for algo in ("auto", "dist", "direct"):
        _check(b, dev, algo)
        if algo != "direct":
            assert _native.fallback_tiles(2, 64, 128) >= 2 * 64 * 128 // 16, "every pixel block of both items was meant to go direct"
    # a batch of one item with the trend and one without: the decision is per item
    plain = synth.make_batch(9, 2, C=67, D=64, H=64, W=128, V=1, pose=pose)
    mixed = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    for k in ("ref", "src"):
        mixed[k][1] = plain[k][1]
    _check(mixed, dev, "auto")
    assert 64 * 128 // 16 <= _native.fallback_tiles(2, 64, 128) < 2 * 64 * 128 // 16


def test_offset_features_border_cells_and_views(dev):
    """Cells on the image border are where centring is NOT free (a tap outside the image reads zero, not the mean): a pose
    that pushes many samples across the border, two source views, an off-centre principal point, ragged sizes."""
    for H, W, V, pose, D in ((37, 83, 2, "wide", 48), (64, 96, 2, "mono", 100), (50, 70, 1, "stereo", 64)):
        b = synth.make_batch(77, 1, C=19, D=D, H=H, W=W, V=V, pose=pose, cx_off=1.7, cy_off=-0.9)
        g = torch.Generator().manual_seed(5)
        mu = (torch.rand(19, generator=g) * 2 - 1) * 6.0
        b["ref"] = b["ref"] + mu[None, :, None, None]
        b["src"] = b["src"] + mu[None, None, :, None, None]
        ocost, ologp, odepth = oracle_batch(b)
        d = to_dev(b, dev)
        for algo in ("auto",):
            cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0,
                                              algo=algo, want_cost=True)
            fin = torch.isfinite(ocost)
            assert torch.equal(torch.isfinite(cost.cpu()), fin), (algo, pose)
            np.testing.assert_allclose(cost.cpu()[fin].numpy(), ocost[fin].numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=f"{algo} {pose}")
            dfin = torch.isfinite(odepth)
            assert (depth.cpu()[dfin] - odepth[dfin]).abs().max().item() <= DEPTH_ATOL * max(1.0, float(np.max(b["d_candi"])) / 40.0)


def test_centred_features_do_not_raise_the_guard(dev):
    """N(0,1) features: the tiled kernel keeps its correlation-form plane group (the guard costs the headline nothing)."""
    b = synth.make_batch(3, 2, C=67, D=64, H=64, W=128, V=1, pose="mono")
    _check(b, dev, "tiled1")
    assert _native.noncentred_guard(2, 64, 128) is False


def _same_answer(algo, packed, nchw):
    """The packed entry against the NCHW entry of the same kernel.  Bit for bit for the float4 layouts; the distance-form
    kernel's NCHW entry (round 6) takes its channel statistics over the source views AND the reference view, which
    pdepth_pack_source_f32 does not have: another centring constant and scale, the same costs to rounding."""
    for p_, n_ in zip(packed, nchw):
        if algo in ("auto", "dist"):
            assert torch.allclose(p_, n_, rtol=2e-5, atol=5e-5), algo
        else:
            assert torch.equal(p_, n_), algo


def test_packed_source_is_tied_to_the_kernel_family(dev):
    """The centred layout (ALGO_AUTO, L2) and the plain one (the tiled kernel: L1, forced selectors) differ, and
    the library cannot tell them apart from the host: the binding refuses to sweep a packed source with a descriptor that
    selects the other family (pdepth_sweep_centres_source), and both families give the NCHW entry's answer bit for bit."""
    b = _offset_batch("uniform8", "mono", B=1)
    d = to_dev(b, dev)
    args = (d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    centred = ops.pack_source(d["src"], 64)                      # auto, L2: the distance-form kernel's fp16 planes
    plain = ops.pack_source(d["src"], 64, algo="tiled1")
    assert centred.centred and not plain.centred
    assert (centred.layout, plain.layout) == (_native.LAYOUT_DIST16, _native.LAYOUT_C4)
    for ps, kw in ((centred, dict(algo="tiled1")), (plain, dict(algo="auto")), (centred, dict(feat_dist="L1"))):
        with pytest.raises(RuntimeError, match="another kernel family"):
            ops.sweep_dpv(d["ref"], ps, *args, **kw)
    with pytest.raises(RuntimeError, match="lab builds only"):   # (round 4's correlation-form kernel: make LAB=1)
        ops.sweep_dpv(d["ref"], d["src"], *args, algo="corr")
    first = {}
    for ps, algo in ((centred, "auto"), (plain, "tiled1")):
        cp, lp, dp = ops.sweep_dpv(d["ref"], ps, *args, algo=algo, want_cost=True)
        ca, la, da = ops.sweep_dpv(d["ref"], d["src"], *args, algo=algo, want_cost=True)
        _same_answer(algo, (cp, lp, dp), (ca, la, da))
        first[algo] = cp
    # a second and a third sweep of the same packed source (the kernel leaves the workspace ready for the next call)
    for _ in range(2):
        c2 = ops.sweep_dpv(d["ref"], centred, *args, algo="auto", want_cost=True)[0]
        assert torch.equal(c2, first["auto"])


def test_a_foreign_packed_layout_is_detected_on_the_device(dev):
    """A caller of the C ABI (no binding in between) that hands pdepth_sweep_dpv_packed_f32 a workspace packed for another
    kernel family: pdepth_pack_* writes the layout's tag into the workspace, every packed sweep kernel compares it with the
    layout it reads and, on a mismatch, fills its outputs with NaN instead of interpreting foreign bytes.
    The binding's host-side check is bypassed here by forging the tag it keeps."""
    b = synth.make_batch(5, 1, C=67, D=64, H=64, W=128, V=1, pose="mono")
    d = to_dev(b, dev)
    args = (d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    for pack_algo, sweep_algo in (("tiled1", "auto"), ("auto", "tiled1")):
        ps = ops.pack_source(d["src"], 64, algo=pack_algo)
        honest = ps.layout
        ps.layout = ops.pack_source(d["src"], 64, algo=sweep_algo).layout      # what the sweep's descriptor expects
        assert ps.layout != honest
        cost, logp, depth = ops.sweep_dpv(d["ref"], ps, *args, algo=sweep_algo, want_cost=True)
        for name, out in (("cost", cost), ("logp", logp), ("depth", depth)):
            assert bool(torch.isnan(out).all()), f"packed for {pack_algo}, swept as {sweep_algo}: {name} is not all NaN"
        # the workspace is still what it was: the honest sweep gives the NCHW entry's answer
        ps.layout = honest
        cp, _, dp = ops.sweep_dpv(d["ref"], ps, *args, algo=pack_algo, want_cost=True)
        ca, _, da = ops.sweep_dpv(d["ref"], d["src"], *args, algo=pack_algo, want_cost=True)
        _same_answer(pack_algo, (cp, dp), (ca, da))


def test_passes_that_do_not_fit_are_evaluated_directly(dev):
    """A wide-baseline pose: epipolar segments of hundreds of texels, more blocks of X than the kernel's LDS holds -- those
    passes take the direct evaluation inside the same launch (no tile flags, no second kernel) and meet the same bounds;
    the diagnostics counter says that it happened."""
    total = {"auto": 0}
    for seed, (H, W, D, V) in enumerate(((192, 400, 64, 1), (120, 260, 128, 2))):
        b = synth.make_batch(900 + seed, 1, C=35, D=D, H=H, W=W, V=V, pose="wide")
        g = torch.Generator().manual_seed(seed)
        mu = (torch.rand(35, generator=g) * 2 - 1) * 3.0
        b["ref"] = b["ref"] + mu[None, :, None, None]
        b["src"] = b["src"] + mu[None, None, :, None, None]
        ocost, ologp, odepth = oracle_batch(b)
        d = to_dev(b, dev)
        for algo in ("auto",):
            cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0,
                                              algo=algo, want_cost=True)
            total[algo] += _native.fallback_tiles(1, H, W)
            fin = torch.isfinite(ocost)
            assert torch.equal(torch.isfinite(cost.cpu()), fin)
            np.testing.assert_allclose(cost.cpu()[fin].numpy(), ocost[fin].numpy(), rtol=COST_RTOL, atol=COST_ATOL)
            dfin = torch.isfinite(odepth)
            assert (depth.cpu()[dfin] - odepth[dfin]).abs().max().item() <= DEPTH_ATOL
            # cost only (no softmax epilogue between consecutive pixel blocks)
            c2 = ops.sweep_cost(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0, algo=algo)
            assert torch.equal(c2.nan_to_num(nan=-7.0), cost.nan_to_num(nan=-7.0))
    assert min(total.values()) > 0, f"these poses are meant to exceed the row tables: {total}"


def test_extremes_of_the_default_kernel(dev):
    """The corners of the shape range sweep_dist_supports() admits -- one channel, one plane, images smaller than a tile, the
    widest features (C = 72: 18 packed planes), D = 65 (a second plane group with one plane in it), 8 source views, more than
    64 batch items (the per-item block-shape table holds 64) -- against the oracle, with offset features; and just beyond
    them (9 views, C = 73, D = 129) `auto` still answers (the LDS-tiled kernel) while the forced selector says no."""
    cases = [dict(B=1, C=1, D=1, H=3, W=5, V=1), dict(B=2, C=72, D=65, H=9, W=33, V=8), dict(B=70, C=5, D=7, H=4, W=16, V=1),
             dict(B=1, C=4, D=64, H=1, W=130, V=2), dict(B=1, C=67, D=128, H=21, W=19, V=3)]
    for i, c in enumerate(cases):
        b = synth.make_batch(300 + i, c["B"], C=c["C"], D=c["D"], H=c["H"], W=c["W"], V=c["V"], pose=("mono", "stereo", "wide")[i % 3])
        g = torch.Generator().manual_seed(i)
        mu = (torch.rand(c["C"], generator=g) * 2 - 1) * 4.0
        b["ref"] = b["ref"] + mu[None, :, None, None]
        b["src"] = b["src"] + mu[None, None, :, None, None]
        ocost, ologp, odepth = oracle_batch(b)
        d = to_dev(b, dev)
        for algo in ("auto", "dist"):
            cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0,
                                              algo=algo, want_cost=True)
            fin = torch.isfinite(ocost)
            assert torch.equal(torch.isfinite(cost.cpu()), fin), (algo, c)
            # (the north-star bounds as they stand, whatever the number of views: no scaling with V)
            np.testing.assert_allclose(cost.cpu()[fin].numpy(), ocost[fin].numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=f"{algo} {c}")
            # depth: the plain 1e-4 m.  (The 8-view case -- costs of 8 views x 72 channels with means of 4 sigma: ~600 -- sat at
            # 1.1e-4 m at one pixel in round 5, where the float32 oracle itself is that far from the exact value; the kernel now
            # finds such an item ill-conditioned and hands it to the gather kernel: csrc/sweep_dist.hip, "Conditioning")
            dfin = torch.isfinite(odepth)
            assert float((depth.cpu() - odepth)[dfin].abs().max()) <= DEPTH_ATOL, (algo, c)
    for c in (dict(C=8, D=16, V=9), dict(C=73, D=16, V=1), dict(C=8, D=129, V=1)):
        b = synth.make_batch(400, 1, C=c["C"], D=c["D"], H=12, W=20, V=c["V"], pose="mono")
        ocost, ologp, odepth = oracle_batch(b)
        d = to_dev(b, dev)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
        cost = ops.sweep_dpv(*args, algo="auto", want_cost=True)[0]
        fin = torch.isfinite(ocost)
        np.testing.assert_allclose(cost.cpu()[fin].numpy(), ocost[fin].numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=str(c))
        for algo in ("dist",):
            with pytest.raises(RuntimeError):
                ops.sweep_dpv(*args, algo=algo)
