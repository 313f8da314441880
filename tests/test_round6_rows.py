"""GPU suite, round 6 (VERDICT r5 items 2 and ADVICE r5): what the statistics of the default kernel see, and what it does with
values that are not numbers.

The reference evaluates cost = sum_c (sum_t w_t s_t - r)^2 / sigma in fp32 on whatever it is given
(warping/homography.py:80-82,129) and log_softmax / the expectation propagate a NaN of one plane to the whole pixel
(models/packnet.py:394, utils/img_utils.py:52-61).  The default kernel (csrc/sweep_dist.hip) centres, scales and splits
the features into fp16 pairs from channel statistics (csrc/sweep_pack.hip): the cases below are the ones those statistics
could not see in round 5 (source view 0 only, 8 sampled rows), every one against the CPU oracle at the north-star bound."""
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


def _run(b, dev, algo, sigma=10.0):
    d = to_dev(b, dev)
    cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], sigma, algo=algo,
                                      want_cost=True)
    return cost.cpu(), logp.cpu(), depth.cpu()


def _against_oracle(b, dev, algos=("auto", "direct"), depth_atol=DEPTH_ATOL):
    ocost, ologp, odepth = oracle_batch(b)
    worst = {}
    for algo in algos:
        cost, logp, depth = _run(b, dev, algo)
        # (non-finite where the reference is non-finite: an infinite feature gives the reference an infinite cost, the default
        #  kernel a NaN -- loud either way; depth and log-DPV are NaN in both)
        assert torch.equal(torch.isfinite(cost), torch.isfinite(ocost)), f"{algo}: non-finite pattern of the cost differs from the reference's"
        assert torch.equal(torch.isnan(depth), torch.isnan(odepth)), f"{algo}: NaN pattern of the depth differs from the reference's"
        assert torch.equal(torch.isfinite(logp), torch.isfinite(ologp)), f"{algo}: non-finite pattern of the log-DPV differs from the reference's"
        fin = torch.isfinite(ocost)
        np.testing.assert_allclose(cost[fin].numpy(), ocost[fin].numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=algo)
        dfin = torch.isfinite(odepth)
        err = float((depth - odepth)[dfin].abs().max())
        assert err <= depth_atol, f"{algo}: depth differs from the oracle by {err:.3e} m"
        worst[algo] = err
    return worst


@pytest.mark.parametrize("pose", ["mono", "stereo"])
def test_reference_and_source_with_different_channel_means(dev, pose):
    """An exposure change: the reference view's channels sit up to +-2 sigma from the source's.  The pooled statistics (all
    source views and the reference view) centre between the two; the residual offsets are energy the guard sees."""
    b = synth.make_batch(31, 2, C=67, D=64, H=64, W=128, V=1, pose=pose)
    g = torch.Generator().manual_seed(77)
    dmu = (torch.rand(67, generator=g) * 2 - 1) * 2.0
    b["ref"] = b["ref"] + dmu[None, :, None, None]
    # (costs of ~20 per plane; measured: auto 3.1e-5 / 4.4e-5 m, direct 2.3e-5 / 1.9e-5 m -- the plain north-star bound)
    ocost, _, odepth = oracle_batch(b)
    for algo in ("auto", "direct"):
        cost, _, depth = _run(b, dev, algo)
        np.testing.assert_allclose(cost.numpy(), ocost.numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=algo)
        assert float((depth - odepth).abs().max()) <= DEPTH_ATOL, algo


def test_a_source_view_unlike_view_zero(dev):
    """View 1 fifty times view 0's amplitude: sampled from view 0 alone (round 5) the fp16 scale put view 1 out of range and the
    whole item came out NaN; the pooled maximum covers it.  Costs are of the order of 50^2 C: tolerances relative."""
    b = synth.make_batch(32, 1, C=67, D=32, H=48, W=96, V=2, pose="mono")
    b["src"][:, 1] = b["src"][:, 1] * 50.0
    ocost, _, odepth = oracle_batch(b)
    for algo in ("auto", "direct"):
        cost, _, depth = _run(b, dev, algo)
        assert torch.isfinite(cost).all() and torch.isfinite(depth).all(), algo
        np.testing.assert_allclose(cost.numpy(), ocost.numpy(), rtol=3e-5, atol=1e-3, err_msg=algo)


def test_a_bright_row_the_statistics_do_not_sample(dev):
    """One row of the source at 1e6 where everything else is N(0, 1): the sampled maximum misses it, the pack finds it out of the
    fp16 range.  Never a wrong number: on the NCHW entry the item goes to the gather kernel and comes out RIGHT (round 6 routing;
    ADVICE r5: "evaluate the item in the fp32 reference form from a.src"), through a packed workspace -- no NCHW source to fall
    back to -- it comes out NaN; the other batch item is untouched."""
    b = synth.make_batch(33, 2, C=67, D=32, H=64, W=128, V=1, pose="mono")
    rows = {0, 4, 12, 20, 28, 36, 44, 52, 60}   # (any row but the sampled ones: stats_row(i, 8, 64) = 4, 12, .., 60 -- and their pooled variants)
    row = next(r for r in range(1, 63) if r not in rows and r % 4 != 0)
    b["src"][1, 0, :, row, :] = 1.0e6
    ocost, _, odepth = oracle_batch(b)
    cost, _, depth = _run(b, dev, "auto")
    assert torch.isfinite(cost[0]).all() and float((depth[0] - odepth[0]).abs().max()) <= DEPTH_ATOL
    np.testing.assert_allclose(cost[1].numpy(), ocost[1].numpy(), rtol=3e-5, atol=1e-2)
    assert float((depth[1] - odepth[1]).abs().max()) <= DEPTH_ATOL
    d = to_dev(b, dev)
    packed = ops.pack_source(d["src"], 32, "auto")
    pcost, _, pdepth = ops.sweep_dpv(d["ref"], packed, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0, want_cost=True)
    assert torch.isfinite(pcost[0]).all() and torch.isnan(pcost[1]).all() and torch.isnan(pdepth[1]).all()
    cost, _, depth = _run(b, dev, "direct")
    np.testing.assert_allclose(cost.numpy(), ocost.numpy(), rtol=3e-5, atol=1e-2)


@pytest.mark.parametrize("pose", ["mono", "stereo"])
def test_a_nan_reference_pixel_and_a_non_finite_plane(dev, pose):
    """exp(max(x, -1000)) and a clamp of the reference features swallowed NaN.  One reference pixel NaN in one
    channel, one +inf: every cost, log-probability and the depth of exactly those pixels is NaN, as in the reference."""
    b = synth.make_batch(34, 1, C=67, D=64, H=48, W=96, V=1, pose=pose)
    b["ref"][0, 5, 10, 20] = float("nan")
    b["ref"][0, 66, 30, 7] = float("inf")
    ocost, ologp, odepth = oracle_batch(b)
    assert torch.isnan(odepth[0, 10, 20]) and torch.isnan(odepth[0, 30, 7]) and int(torch.isnan(odepth).sum()) == 2
    _against_oracle(b, dev)


def test_a_nan_candidate_plane(dev):
    """A NaN depth candidate: its plane's positions are NaN, the reference's cost there is |r|^2-like with NaN weights -> NaN; the
    softmax over the planes turns every pixel's log-DPV and depth NaN.  The kernels agree with the oracle's pattern."""
    b = synth.make_batch(35, 1, C=67, D=16, H=32, W=64, V=1, pose="mono")
    dc = np.array(b["d_candi"], dtype=np.float32).copy()
    dc[7] = np.nan
    b["d_candi"] = dc
    ocost, ologp, odepth = oracle_batch(b)
    for algo in ("auto", "direct"):
        cost, logp, depth = _run(b, dev, algo)
        assert torch.equal(torch.isfinite(cost), torch.isfinite(ocost)), algo
        assert torch.equal(torch.isnan(depth), torch.isnan(odepth)), algo
        fin = torch.isfinite(ocost)
        np.testing.assert_allclose(cost[fin].numpy(), ocost[fin].numpy(), rtol=COST_RTOL, atol=COST_ATOL, err_msg=algo)


def test_routed_items_are_the_gather_kernels_bit_for_bit(dev):
    """Round 6 routing (csrc/sweep_dist.hip: "Conditioning"): in a batch of three, item 1 carries per-channel offsets of 6 sigma
    (costs of hundreds: ill-conditioned by the kernel's measure), items 0 and 2 are the usual N(0, 1).  The default selector gives
    item 1 the gather kernel's answer bit for bit -- it IS the gather kernel's --, items 0 and 2 the fast form's (what they get in
    a batch without item 1), and says so in its diagnostics.  Every item within the north star of the oracle."""
    b = synth.make_batch(52, 3, C=67, D=64, H=48, W=96, V=2, pose="mono")
    g = torch.Generator().manual_seed(9)
    mu = (torch.rand(67, generator=g) * 2 - 1) * 6.0
    b["ref"][1] += mu[:, None, None]
    b["src"][1] += mu[None, :, None, None]
    ocost, _, odepth = oracle_batch(b)
    cost, logp, depth = _run(b, dev, "auto")
    routed = _native.fallback_tiles(3, 48, 96)
    dcost, dlogp, ddepth = _run(b, dev, "direct")
    assert routed == 48 * 96 // 16, routed            # (every pixel block of item 1, none of the others)
    assert torch.equal(cost[1], dcost[1]) and torch.equal(logp[1], dlogp[1]) and torch.equal(depth[1], ddepth[1])
    assert not torch.equal(cost[0], dcost[0])         # (the fast form: another rounding)
    two = {k: (v[[0, 2]] if isinstance(v, torch.Tensor) and v.shape[:1] == (3,) else v) for k, v in b.items()}
    cost2, _, depth2 = _run(two, dev, "auto")
    assert torch.equal(cost[[0, 2]], cost2) and torch.equal(depth[[0, 2]], depth2)
    assert float((depth - odepth).abs().max()) <= DEPTH_ATOL


def test_nchw_entry_is_capturable(dev):
    """the NCHW call of the default kernel -- statistics, pack, sweep: no host synchronisation, no allocation --
    captured into a graph and replayed on NEW features in the captured buffers gives the eager call's answer bit for bit.  (Also
    with -DPDEPTH_PACK_FUSE_STATS=1, the statistics inside the pack kernel: their tags are a kernel argument a replay repeats, the
    sweep kernel clears them.  That build was 2-4 us per call SLOWER on every shape -- csrc/pack_dist.hip -- and is off.)"""
    b = synth.make_batch(41, 2, C=67, D=64, H=64, W=128, V=1, pose="mono")
    d = to_dev(b, dev)
    ref, src = d["ref"].clone(), d["src"].clone()
    run = lambda: ops.sweep_dpv(ref, src, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    want0 = [x.clone() for x in run()[1:]]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = run()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[1], want0[0]) and torch.equal(out[2], want0[1])
    for seed, scale in ((42, 1.0), (43, 30.0), (44, 0.02)):     # other frames, other magnitudes: another scale exponent every time
        b2 = synth.make_batch(seed, 2, C=67, D=64, H=64, W=128, V=1, pose="mono")
        ref.copy_(b2["ref"].to(dev) * scale); src.copy_(b2["src"].to(dev) * scale)
        graph.replay()
        torch.cuda.synchronize()
        got = [out[1].clone(), out[2].clone()]
        want = run()
        assert torch.equal(got[0], want[1]) and torch.equal(got[1], want[2]), (seed, scale)
