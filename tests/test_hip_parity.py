"""GPU suite: the HIP path (through the C ABI) against the golden fixtures and the CPU oracle.

Tolerances
  sample coordinates : bit-exact (integer-like contract, see csrc/geometry.hpp)
  cost / logp        : 2e-4 abs + 2e-5 rel (fp32 summation order over C differs from ATen's cascade sum)
  depth              : 1e-4 abs  (BASELINE.json north_star)
"""
import numpy as np
import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import _native, ops, synth
from pdepth_amd.utils import img_utils
from pdepth_amd.warping import homography
from oracle import ref_cpu as O
from util import DEPTH_ATOL, golden, golden_blas, oracle_batch, to_dev

pytestmark = pytest.mark.gpu

COST_ATOL, COST_RTOL = 2e-4, 2e-5


def _cam(g, dev):
    K = torch.from_numpy(g["K"]).to(dev)
    return {"intrinsic_M_cuda": K, "intrinsic_M": g["K"], "unit_ray_array_2D": torch.from_numpy(g["rays"]).to(dev)}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU suite needs a GPU"
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------------------------
# golden fixtures through the reference-compatible per-item API
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g1_rot_trans", "g2_identity", "g3_out_of_bounds"])
@pytest.mark.parametrize("metric", ["L2", "L1"])
def test_golden_tiny_sweeps(dev, name, metric):
    g = golden(name + ".npz")
    cost = homography.est_swp_volume_v4(torch.from_numpy(g["ref"]).to(dev), torch.from_numpy(g["src"]).to(dev),
                                        g["d_candi"], torch.from_numpy(g["R"]).to(dev),
                                        torch.from_numpy(g["t"]).to(dev), _cam(g, dev), float(g["sigma"]),
                                        feat_dist=metric, blas=golden_blas(g))
    assert cost.shape == (1, 8, 16, 24) and cost.device.type == "cuda"
    np.testing.assert_allclose(cost.cpu().numpy(), g["cost_" + metric], rtol=COST_RTOL, atol=COST_ATOL)


@pytest.mark.parametrize("name", ["g4_stereo_64x96", "g4_mono_64x128"])
def test_golden_model_real(dev, name):
    g = golden(name + ".npz")
    kw = eval(str(g["synth_kwargs"]), {"__builtins__": {}}, {"dict": dict})
    it = synth.make_item(**kw)
    b = to_dev({k: (v[None] if isinstance(v, torch.Tensor) else v) for k, v in it.items()}, dev)
    cost, logp, depth = ops.sweep_dpv(b["ref"], b["src"], b["K"], b["R"], b["t"], b["rays"], b["cxcy"],
                                      b["d_candi"], 10.0, want_cost=True, blas=golden_blas(g))
    np.testing.assert_allclose(cost.cpu().numpy()[:, ::4, ::2, ::2], g["cost_sub"], rtol=COST_RTOL, atol=COST_ATOL)
    np.testing.assert_allclose(logp.cpu().numpy()[:, ::4, ::2, ::2], g["logp_sub"], rtol=COST_RTOL, atol=COST_ATOL)
    err = np.abs(depth.cpu().numpy() - g["depth"]).max()
    assert err <= DEPTH_ATOL, f"depth differs from the reference fixture by {err:.3e}"


def test_golden_dpv(dev):
    g = golden("g5_dpv.npz")
    for nm in ("broad", "peaked"):
        x = torch.from_numpy(g[nm + "_logits"]).to(dev)
        logp, depth = ops.dpv_reduce(x, g["d_candi"])
        np.testing.assert_allclose(logp.cpu().numpy(), g[nm + "_logp"], rtol=1e-6, atol=2e-6)
        assert np.abs(depth.cpu().numpy() - g[nm + "_depth_log"]).max() <= DEPTH_ATOL
        d1 = img_utils.dpv_to_depthmap(torch.from_numpy(g[nm + "_logp"]).to(dev), g["d_candi"], BV_log=True)
        assert np.abs(d1.cpu().numpy() - g[nm + "_depth_log"]).max() <= DEPTH_ATOL
        d2 = img_utils.dpv_to_depthmap(torch.exp(torch.from_numpy(g[nm + "_logp"])).to(dev), g["d_candi"])
        assert np.abs(d2.cpu().numpy() - g[nm + "_depth_lin"]).max() <= DEPTH_ATOL
    with pytest.raises(Exception, match="Unable to handle this case"):
        img_utils.dpv_to_depthmap(torch.zeros(2, 4, 3, 3, device=dev), g["d_candi"][:4])


def test_golden_warp_feature(dev):
    g = golden("g6_warp_feature.npz")
    out = homography.warp_feature(torch.from_numpy(g["feat"]).to(dev), g["d_candi"], torch.from_numpy(g["R"]).to(dev),
                                  torch.from_numpy(g["t"]).to(dev), _cam(g, dev), blas=golden_blas(g))
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-6, atol=1e-6)
    with pytest.raises(Exception, match="Warped Accum Error"):
        homography.warp_feature(torch.zeros(2, 1, 8, 4, 4, device=dev), g["d_candi"], None, None, _cam(g, dev))


# ---------------------------------------------------------------------------------------------
# seeded inputs against the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pose,H,W,off", [("mono", 60, 100, 1.3), ("stereo", 64, 128, 0.0), ("mono", 33, 47, -0.6)])
def test_sample_coordinates_bit_exact(dev, pose, H, W, off):
    b = synth.make_batch(21, 2, C=1, D=16, H=H, W=W, V=2, pose=pose, cx_off=off, cy_off=-off / 2)
    d = to_dev(b, dev)
    ix, iy = ops.sample_coords(d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], H, W)
    for bi in range(2):
        for v in range(2):
            K = b["K"][bi]
            ox, oy = O.sample_coords(K, b["R"][bi, v], b["t"][bi, v], b["rays"][bi], b["d_candi"],
                                     K.numpy()[0, 2], K.numpy()[1, 2], H, W)
            gx, gy = ix[bi, v].cpu().reshape(16, -1), iy[bi, v].cpu().reshape(16, -1)
            bad = int((gx != ox).sum() + (gy != oy).sum())
            assert bad == 0, f"{bad} of {2 * gx.numel()} sample coordinates differ from the CPU path"


@pytest.mark.parametrize("cfg", [
    dict(C=67, D=64, H=64, W=128, V=1, pose="mono"),
    dict(C=67, D=64, H=64, W=128, V=1, pose="stereo"),
    dict(C=7, D=8, H=17, W=23, V=3, pose="mono", cx_off=1.1, cy_off=-0.4),   # ragged sizes, 3 views
    dict(C=70, D=16, H=20, W=36, V=2, pose="mono"),                          # C > 68: chunked-channel kernel
    dict(C=67, D=128, H=16, W=64, V=1, pose="stereo"),                       # D=128 (config 5 depth count)
    dict(C=64, D=64, H=32, W=64, V=1, pose="mono", peaked=True),
    dict(C=6, D=24, H=96, W=320, V=1, pose="wide"),                          # large disparity per plane: split windows
])
@pytest.mark.parametrize("metric", ["L2", "L1"])
@pytest.mark.parametrize("algo", ["auto", "direct"])
def test_sweep_matches_oracle(dev, cfg, metric, algo):
    if metric == "L1" and cfg["D"] == 128:
        pytest.skip("covered by L2")
    b = synth.make_batch(31, 2, **cfg)
    ocost, ologp, odepth = _oracle_cached(repr(sorted(cfg.items())), metric, b)
    d = to_dev(b, dev)
    cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"],
                                      d["d_candi"], 10.0, feat_dist=metric, want_cost=True, algo=algo)
    np.testing.assert_allclose(cost.cpu().numpy(), ocost.numpy(), rtol=COST_RTOL, atol=COST_ATOL)
    np.testing.assert_allclose(logp.cpu().numpy(), ologp.numpy(), rtol=COST_RTOL, atol=COST_ATOL)
    err = (depth.cpu() - odepth).abs().max().item()
    assert err <= DEPTH_ATOL, f"depth differs from the CPU oracle by {err:.3e}"
    # cost-only entry point returns the same volume
    cost2 = ops.sweep_cost(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0,
                           feat_dist=metric, algo=algo)
    assert torch.equal(cost, cost2)
    # depth-only request (tiled kernel then keeps its costs in workspace scratch)
    _, _, depth3 = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"],
                                 10.0, feat_dist=metric, want_cost=False, want_logp=False, want_depth=True, algo=algo)
    assert torch.equal(depth, depth3)


_ORACLE_CACHE = {}


def _oracle_cached(key, metric, batch):
    if (key, metric) not in _ORACLE_CACHE:
        _ORACLE_CACHE[(key, metric)] = oracle_batch(batch, metric)
    return _ORACLE_CACHE[(key, metric)]


@pytest.mark.parametrize("pose_kind", ["big_rotation", "sideways", "behind"])
def test_tiled_falls_back_to_gather_on_extreme_poses(dev, pose_kind):
    """Windows that do not fit LDS (or samples behind the camera) take the gather kernel tile by tile;
    the result must equal the all-gather path (same arithmetic order per pixel)."""
    b = synth.make_batch(41, 2, C=12, D=16, H=40, W=72, V=2, pose="mono")
    R = b["R"].clone(); t = b["t"].clone()
    if pose_kind == "big_rotation":
        c, s_ = np.cos(0.6), np.sin(0.6)
        R[:, 0] = torch.tensor([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], dtype=torch.float32)  # 34 deg roll
    elif pose_kind == "sideways":
        t[:, 0] = torch.tensor([6.0, 2.0, 0.0])
    else:
        t[:, 1] = torch.tensor([0.0, 0.0, -20.0])  # some planes end up behind the source camera
    b["R"], b["t"] = R, t
    d = to_dev(b, dev)
    args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    c_auto, l_auto, d_auto = ops.sweep_dpv(*args, want_cost=True, algo="auto")
    c_dir, l_dir, d_dir = ops.sweep_dpv(*args, want_cost=True, algo="direct")
    np.testing.assert_allclose(c_auto.cpu().numpy(), c_dir.cpu().numpy(), rtol=1e-5, atol=1e-5, equal_nan=True)
    np.testing.assert_allclose(d_auto.cpu().numpy(), d_dir.cpu().numpy(), rtol=0, atol=DEPTH_ATOL, equal_nan=True)
    ocost, _, odepth = oracle_batch(b)
    np.testing.assert_allclose(c_auto.cpu().numpy(), ocost.numpy(), rtol=COST_RTOL, atol=COST_ATOL, equal_nan=True)


def test_strided_views_no_copy(dev):
    """ref = last view, sources = leading views of ONE [B,V+1,C,H,W] tensor (models/models.py:533-534)."""
    b = synth.make_batch(32, 2, C=9, D=8, H=12, W=20, V=2, pose="mono")
    allv = torch.cat([b["src"], b["ref"][:, None]], dim=1).to(dev)
    d = to_dev(b, dev)
    want = ops.sweep_cost(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    got = ops.sweep_cost(allv[:, -1], allv[:, :-1], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    assert torch.equal(want, got)


def test_dpv_reduce_shapes_and_properties(dev):
    rng = np.random.default_rng(5)
    for (B, D, H, W) in ((2, 64, 32, 48), (1, 128, 8, 12), (3, 7, 5, 9), (1, 200, 4, 6), (2, 64, 3, 5)):
        x = torch.from_numpy(rng.standard_normal((B, D, H, W), dtype=np.float32) * 3)
        dc = O.powerf(5.0, 40.0, D, 1.0)
        logp, depth = ops.dpv_reduce(x.to(dev), dc)
        want_lp = O.log_dpv(x)
        np.testing.assert_allclose(logp.cpu().numpy(), want_lp.numpy(), rtol=1e-6, atol=3e-6)
        for bi in range(B):
            want_d = O.dpv_to_depthmap(want_lp[bi:bi + 1], dc, BV_log=True)
            assert (depth[bi:bi + 1].cpu() - want_d).abs().max().item() <= DEPTH_ATOL
        # properties: probabilities sum to one, depth inside [d_min, d_max], shift invariance
        assert (torch.exp(logp).sum(1) - 1).abs().max().item() < 1e-5
        assert depth.min().item() >= 5.0 - 1e-4 and depth.max().item() <= 40.0 + 1e-4
        logp2, depth2 = ops.dpv_reduce((x + 3.0).to(dev), dc)
        assert (logp2 - logp).abs().max().item() < 1e-5 and (depth2 - depth).abs().max().item() < 1e-4
        # in-place variant
        xi = x.to(dev).clone()
        lp3, _ = ops.dpv_reduce(xi, dc, inplace=True)
        assert lp3.data_ptr() == xi.data_ptr() and torch.equal(lp3, logp)


def test_full_size_properties(dev):
    """BASELINE config 2 size (B=4, D=64, 256x512, C=67): properties that need no oracle."""
    b = synth.make_batch(2, 2, C=67, D=64, H=256, W=512, V=1, pose="stereo")
    d = to_dev(b, dev)
    cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"],
                                      d["d_candi"], 10.0, want_cost=True)
    assert torch.isfinite(cost).all() and (cost >= 0).all()
    assert (torch.exp(logp).sum(1) - 1).abs().max().item() < 1e-5
    assert depth.min().item() >= 5.0 - 1e-4 and depth.max().item() <= 40.0 + 1e-4
    # fused outputs == unfused chain on the same cost volume
    lp2, dp2 = ops.dpv_reduce(cost, d["d_candi"])
    assert (lp2 - logp).abs().max().item() < 2e-5 and (dp2 - depth).abs().max().item() < DEPTH_ATOL
    # identity pose with src == ref gives ~zero cost everywhere (linearity of the sampler)
    eye = torch.eye(3, device=dev).reshape(1, 1, 3, 3).repeat(2, 1, 1, 1)
    zero = torch.zeros(2, 1, 3, device=dev)
    c0 = ops.sweep_cost(d["ref"], d["ref"][:, None], d["K"], eye, zero, d["rays"], d["cxcy"], d["d_candi"], 10.0)
    assert c0.abs().max().item() < 1e-4
    # a few oracle spot-rows at full size would take minutes; one 8-row strip through the oracle instead:
    strip = {k: v for k, v in b.items()}
    oc, _, _ = oracle_batch({**strip, "ref": b["ref"][:1], "src": b["src"][:1], "K": b["K"][:1], "R": b["R"][:1],
                             "t": b["t"][:1], "rays": b["rays"][:1], "cxcy": b["cxcy"][:1]})
    np.testing.assert_allclose(cost[:1].cpu().numpy(), oc.numpy(), rtol=COST_RTOL, atol=COST_ATOL)


@pytest.mark.parametrize("cfg", [
    dict(C=1, D=1, H=1, W=1, V=1),      # smallest possible problem
    dict(C=3, D=2, H=1, W=37, V=1),     # a single row
    dict(C=3, D=5, H=29, W=1, V=2),     # a single column
    dict(C=2, D=160, H=6, W=20, V=1),   # largest D of the LDS-tiled kernel
    dict(C=2, D=161, H=6, W=20, V=1),   # one more: AUTO switches to the gather kernel
    dict(C=2, D=512, H=4, W=16, V=1),   # largest D of the gather kernel
])
def test_extreme_shapes(dev, cfg):
    b = synth.make_batch(51, 1, pose="mono", **cfg)
    ocost, ologp, odepth = oracle_batch(b)
    d = to_dev(b, dev)
    for algo in ("auto", "direct"):
        cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"],
                                          d["d_candi"], 10.0, want_cost=True, algo=algo)
        np.testing.assert_allclose(cost.cpu().numpy(), ocost.numpy(), rtol=COST_RTOL, atol=COST_ATOL)
        assert (depth.cpu() - odepth).abs().max().item() <= DEPTH_ATOL


def test_argument_errors(dev):
    b = to_dev(synth.make_batch(52, 1, C=4, D=8, H=8, W=8, V=1), dev)
    args = [b["ref"], b["src"], b["K"], b["R"], b["t"], b["rays"], b["cxcy"], b["d_candi"], 10.0]
    with pytest.raises(Exception, match="undefined metric"):
        ops.sweep_cost(*args, feat_dist="cosine")
    with pytest.raises(RuntimeError, match="exceeds"):
        ops.sweep_cost(*args[:7], synth.powerf(5, 40, 513, 1.0), 10.0)
    with pytest.raises(RuntimeError, match="expected float32"):
        ops.sweep_cost(b["ref"].double(), *args[1:])
    with pytest.raises(RuntimeError, match="does not match"):
        ops.sweep_cost(b["ref"], b["src"][:, :, :3], *args[2:])
    with pytest.raises(RuntimeError, match="expected shape"):
        ops.sweep_cost(b["ref"], b["src"], b["K"][:, :2], *args[3:])
    with pytest.raises(RuntimeError, match="sigma"):
        ops.sweep_cost(*args[:8], 0.0)
    # non-contiguous inner layout is copied, not rejected
    ref_t = b["ref"].permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)
    assert not ref_t.is_contiguous()
    assert torch.equal(ops.sweep_cost(ref_t, *args[1:]), ops.sweep_cost(*args))


def test_fast_divide_chain_is_bit_identical_to_ieee(dev):
    """The LDS-tiled kernel computes sample positions with an explicit fma divide chain that shares reciprocals
    (geometry.hpp, plane_sample_pos_fast); the gather kernel lets the compiler emit IEEE divides.  Both are exposed
    through pdepth_sample_coords_f32 (algo AUTO / DIRECT) and must agree bit for bit wherever the position is
    finite and anywhere near the image -- over ordinary, strongly rotated, behind-the-camera and huge-baseline
    poses, odd principal points and clustered depth candidates (about 10 M positions)."""
    rng = np.random.default_rng(77)
    total = 0
    for case in range(60):
        H, W = int(rng.integers(8, 90)), int(rng.integers(8, 160))
        D, V = int(rng.integers(4, 96)), int(rng.integers(1, 4))
        b = synth.make_batch(900 + case, 1, C=1, D=D, H=H, W=W, V=V, pose=("mono", "stereo", "wide")[case % 3],
                             cx_off=float(rng.uniform(-3, 3)), cy_off=float(rng.uniform(-2, 2)))
        kind = case % 5
        if kind == 1:
            ang = rng.uniform(-0.4, 0.4)
            cz, sz = np.cos(ang), np.sin(ang)
            b["R"][0, 0] = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float32) @ b["R"][0, 0]
            b["t"][0, 0] = torch.from_numpy(rng.uniform(-3, 3, size=3).astype(np.float32))
        elif kind == 2:
            b["t"][0, 0] = torch.from_numpy(rng.uniform(-40, 40, size=3).astype(np.float32))
        elif kind == 3:
            b["d_candi"] = rng.uniform(0.3, 80.0, size=D)
        elif kind == 4:
            b["t"][0, 0] = torch.from_numpy((rng.uniform(-1, 1, size=3) * 1e-3).astype(np.float32))
        d = to_dev(b, dev)
        args = (d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], H, W)
        fx, fy = ops.sample_coords(*args, algo=_native.ALGO_AUTO)
        ex, ey = ops.sample_coords(*args, algo=_native.ALGO_DIRECT)
        near = (ex.abs() < 1e6) & (ey.abs() < 1e6)   # exponent-range corner cases only exist far outside
        bad = int(((fx != ex) | (fy != ey))[near].sum())
        assert bad == 0, f"case {case}: {bad} of {int(near.sum())} positions differ between the divide variants"
        far = ~near
        # far outside (or NaN): both variants must still classify the sample as out of bounds / NaN alike
        assert bool(((fx.isnan() | (fx.abs() >= 1e6) | (fy.abs() >= 1e6) | fy.isnan())[far]).all())
        total += int(near.sum())
    assert total > 5_000_000


@pytest.mark.parametrize("pose", ["mono", "stereo"])
def test_band_mode_agrees_with_gather_to_an_ulp_of_the_cost(dev, pose):
    """The tiled kernel evaluates far planes in correlation form (w^T G w - 2 w.X + |r|^2).  Its costs must agree
    with the gather kernel (reference op order) to an ulp or two of the largest cost of the volume, whatever the
    feature magnitude -- i.e. it adds no error beyond what re-ordering an fp32 sum adds anyway."""
    for scale in (1.0, 4.0, 16.0):
        b = synth.make_batch(5, 1, C=67, D=64, H=64, W=128, V=1, pose=pose, peaked=True)
        d = to_dev(b, dev)
        args = (d["ref"] * scale, d["src"] * scale, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
        ca = ops.sweep_cost(*args, algo="auto")
        cd = ops.sweep_cost(*args, algo="direct")
        rel = float((ca - cd).abs().max() / cd.abs().max())
        assert rel < 2e-6, f"scale {scale}: tiled vs gather differ by {rel:.2e} of the largest cost"
    assert _native.fallback_tiles(1, 64, 128) == 0


def test_fuzz_tiled_against_gather(dev):
    """Random shapes / poses / intrinsics: the LDS-tiled kernel (with window splitting and per-tile fallback)
    must agree with the gather kernel, which evaluates every pixel independently in the reference's op order.
    Catches windowing / splitting / zero-padding mistakes that the handful of structured cases could miss."""
    rng = np.random.default_rng(2024)
    worst, worst_depth = 0.0, 0.0
    for case in range(120):
        H, W = int(rng.integers(3, 70)), int(rng.integers(3, 110))
        C, D, V = int(rng.integers(1, 12)), int(rng.integers(1, 80)), int(rng.integers(1, 4))
        b = synth.make_batch(60 + case, 1, C=C, D=D, H=H, W=W, V=V, pose="mono",
                             cx_off=float(rng.uniform(-2, 2)), cy_off=float(rng.uniform(-1, 1)))
        kind = case % 4
        if kind == 1:    # strong rotation + sideways motion: slanted epipolar lines, large windows
            ang = rng.uniform(-0.25, 0.25, size=3)
            cz, sz = np.cos(ang[2]), np.sin(ang[2])
            Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float32)
            b["R"][0, 0] = torch.from_numpy(Rz) @ b["R"][0, 0]
            b["t"][0, 0] = torch.from_numpy(rng.uniform(-2.5, 2.5, size=3).astype(np.float32))
        elif kind == 2:  # mostly out of the image / behind the camera
            b["t"][0, 0] = torch.from_numpy(rng.uniform(-30, 30, size=3).astype(np.float32))
        elif kind == 3:  # non-monotonic, clustered depth candidates
            b["d_candi"] = np.sort(rng.uniform(0.5, 60.0, size=D))[::-1].copy() if case % 8 == 3 else rng.uniform(2.0, 50.0, size=D)
        d = to_dev(b, dev)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 7.5)
        ca, la, da = ops.sweep_dpv(*args, want_cost=True, algo="auto")
        cd, ld, dd = ops.sweep_dpv(*args, want_cost=True, algo="direct")
        ca, cd = ca.cpu().numpy(), cd.cpu().numpy()
        assert np.array_equal(np.isnan(ca), np.isnan(cd)), f"case {case}: NaN pattern differs"
        scale = max(1.0, float(np.nanmax(np.abs(cd))) if np.isfinite(cd).any() else 1.0)
        err = float(np.nanmax(np.abs(ca - cd))) / scale if np.isfinite(cd).any() else 0.0
        worst = max(worst, err)
        assert err < 2e-5, f"case {case} (kind {kind}, {H}x{W}, C={C}, D={D}, V={V}): tiled vs gather differ by {err:.3e} (relative)"
        fin = np.isfinite(dd.cpu().numpy())
        if fin.any():
            # the north star's 1e-4 (at the reference's 5..40 m range; scaled with the depth range of the case) between
            # the two kernels as well: measured worst 3.6e-5
            ddiff = np.abs(da.cpu().numpy() - dd.cpu().numpy())[fin].max()
            worst_depth = max(worst_depth, ddiff / max(1.0, float(np.max(b["d_candi"])) / 40.0))
            assert ddiff < 1e-4 * max(1.0, float(np.max(b["d_candi"])) / 40.0), f"case {case}: depth differs by {ddiff:.2e}"
    print(f"fuzz: worst relative cost difference tiled vs gather {worst:.2e}, worst depth difference (40 m scale) {worst_depth:.2e}")


def test_fuzz_large_tiled_against_gather(dev):
    """Larger random shapes (up to 260x520, D up to 130, mono / stereo / wide-baseline poses, unsorted depth
    candidates): every path of the tiled kernel -- direct groups with and without window splits, band groups of all
    sizes, per-tile fallback -- against the gather kernel, to an ulp or two of the largest cost."""
    rng = np.random.default_rng(99)
    worst, fallback = 0.0, 0
    for case in range(40):
        H, W = int(rng.integers(40, 260)), int(rng.integers(60, 520))
        C, D, V = int(rng.integers(1, 70)), int(rng.integers(8, 130)), int(rng.integers(1, 3))
        b = synth.make_batch(300 + case, 1, C=C, D=D, H=H, W=W, V=V, pose=("mono", "stereo", "wide")[case % 3],
                             cx_off=float(rng.uniform(-2, 2)), cy_off=float(rng.uniform(-1, 1)))
        if case % 5 == 4:
            b["d_candi"] = rng.uniform(2.0, 50.0, size=D)
        d = to_dev(b, dev)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 7.5)
        ca = ops.sweep_cost(*args, algo="auto").cpu().numpy()
        fallback += _native.fallback_tiles(1, H, W)
        cd = ops.sweep_cost(*args, algo="direct").cpu().numpy()
        assert np.array_equal(np.isnan(ca), np.isnan(cd)), f"case {case}: NaN pattern differs"
        err = float(np.nanmax(np.abs(ca - cd))) / max(1.0, float(np.nanmax(np.abs(cd))))
        worst = max(worst, err)
        assert err < 2e-6, f"case {case} ({H}x{W}, C={C}, D={D}, V={V}): tiled vs gather differ by {err:.3e} (relative)"
    assert fallback > 0  # the wide-baseline cases must have exercised the per-tile fallback as well
    print(f"large fuzz: worst relative difference {worst:.2e}, {fallback} tiles through the gather kernel")


@pytest.mark.parametrize("variant", ["tiled1", "tiled2", "dist"])
def test_every_sweep_implementation_on_small_ragged_shapes(dev, variant):
    """The library holds the distance-form kernel (what ALGO_AUTO runs for the L2 metric) and two builds of the LDS-tiled
    kernel (one / two 16x4 tiles per block); the implementation selectors force one, so that each meets the ragged sizes,
    odd tile counts, multi-view and D > 64 cases whatever ALGO_AUTO would choose."""
    rng = np.random.default_rng(7)
    for case in range(30):
        H, W = int(rng.integers(3, 90)), int(rng.integers(3, 150))
        C, D, V = int(rng.integers(1, 20)), int(rng.integers(1, 100 if variant != "tiled2" else 65)), int(rng.integers(1, 3))
        b = synth.make_batch(700 + case, 2 if case % 4 == 0 else 1, C=C, D=D, H=H, W=W, V=V,
                             pose=("mono", "stereo", "wide")[case % 3], cx_off=float(rng.uniform(-2, 2)))
        d = to_dev(b, dev)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 9.0)
        ca, la, da = ops.sweep_dpv(*args, want_cost=True, algo=variant)
        cd, ld, dd = ops.sweep_dpv(*args, want_cost=True, algo="direct")
        ca, cd = ca.cpu().numpy(), cd.cpu().numpy()
        assert np.array_equal(np.isnan(ca), np.isnan(cd)), f"case {case}: NaN pattern differs"
        err = float(np.nanmax(np.abs(ca - cd))) / max(1.0, float(np.nanmax(np.abs(cd))))
        assert err < 2e-6, f"variant {variant} case {case} ({H}x{W}, C={C}, D={D}, V={V}): {err:.3e}"
        fin = np.isfinite(dd.cpu().numpy())
        if fin.any():
            ddiff = np.abs(da.cpu().numpy() - dd.cpu().numpy())[fin].max()
            assert ddiff < 1.5e-4 * max(1.0, float(np.max(b["d_candi"])) / 40.0), f"variant {variant} case {case}: depth differs by {ddiff:.2e}"



def test_soak_regressions(dev):
    """Two inputs a longer soak run (tools/soak.py, seed 31) found after the fuzz tests above had passed for two rounds.
    (1) A source view 26 m off to the side: the band decision's box of a tile was 30 042 x 84 665 texels, the product
        wrapped in 32 bits and passed the "at most NX_MAX slots" test.  (2) Depth candidates in no particular order:
        every plane opens a new cell, the fast cell-list kernel's per-quad slot totals overflowed their 5-bit fields before
        the "does it fit" test looked at them."""
    def agree(b, algo):
        d = to_dev(b, dev)
        args = (d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 8.0)
        ca = ops.sweep_cost(*args, algo=algo).cpu().numpy()
        cd = ops.sweep_cost(*args, algo="direct").cpu().numpy()
        assert np.array_equal(np.isnan(ca), np.isnan(cd)), algo
        fin = np.isfinite(cd)
        err = float(np.abs(ca - cd)[fin].max()) / max(1.0, float(np.abs(cd[fin]).max()))
        assert err < 2e-6, f"{algo}: {err:.3e}"
    b = synth.make_batch(5480, 1, C=22, D=83, H=195, W=286, V=3, pose="mono", cx_off=1.999560470167534, cy_off=-0.5343920453361704)
    b["t"][0, 0] = torch.tensor([18.91592254, 18.98302991, -11.80695313])
    for algo in ("auto", "tiled1", "dist"):
        agree(b, algo)
    rng = np.random.default_rng(1301)
    for D in (64, 128):
        b = synth.make_batch(6301, 1, C=33, D=D, H=150, W=302, V=1, pose="stereo")
        b["d_candi"] = rng.uniform(0.5, 60.0, size=D)
        for algo in ("dist", "auto", "tiled2" if D == 64 else "tiled1"):
            agree(b, algo)


def test_packed_entry_with_gather_fallback(dev):
    """pdepth_sweep_dpv_packed_f32 has no NCHW source to give to the gather kernel: tiles handed over (wide baseline,
    huge translation) are evaluated from the packed copy -- same taps, same arithmetic, so the outputs are those of the
    plain entry bit for bit.  (Found by the soak run as a GPU fault: the gather kernel dereferenced the NULL source.)"""
    rng = np.random.default_rng(77)
    fb_total = 0
    for pose, C, D, H, W, V, unsorted in (("wide", 19, 96, 180, 400, 2, False), ("stereo", 33, 64, 150, 302, 1, True),
                                          ("wide", 70, 40, 120, 260, 1, True)):
        b = synth.make_batch(8800 + C, 2, C=C, D=D, H=H, W=W, V=V, pose=pose)
        if unsorted:
            b["d_candi"] = rng.uniform(0.5, 60.0, size=D)
        d = to_dev(b, dev)
        args = (d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 9.0)
        for metric in ("L2", "L1"):
            try:
                ps = ops.pack_source(d["src"], D, feat_dist=metric)
                cp, lp, dp = ops.sweep_dpv(d["ref"], ps, *args, feat_dist=metric, want_cost=True)
            except RuntimeError as e:   # shapes the packed entry declines (e.g. L1 with C > 68)
                assert "packed" in str(e), str(e)
                continue
            fb_total += pdepth_amd._native.fallback_tiles(2, H, W)
            ca, la, da = ops.sweep_dpv(d["ref"], d["src"], *args, feat_dist=metric, want_cost=True)
            # (L1: the float4 layout, bit for bit.  L2: the distance-form kernel's NCHW entry takes its channel statistics over
            #  the source views AND the reference view -- round 6 --, the packed entry over the source views: equal to rounding)
            same = torch.equal if metric == "L1" else (lambda x, y: torch.allclose(x, y, rtol=2e-5, atol=1e-4))
            assert same(cp.nan_to_num(nan=-7.0), ca.nan_to_num(nan=-7.0)), (pose, metric)
            assert same(dp.nan_to_num(nan=-7.0), da.nan_to_num(nan=-7.0)), (pose, metric)
            assert same(lp.nan_to_num(nan=-7.0), la.nan_to_num(nan=-7.0)), (pose, metric)
    assert fb_total > 0, "these cases are meant to exercise the gather fallback"
