"""DPV Bayesian fusion (rank 2), the correlation op forward + backward (rank 3), rank-4 pieces."""
import numpy as np
import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import ops
from pdepth_amd.models.correlation import Correlation
from oracle import ref_cpu as O
from util import golden, same_cpu_as_golden


def test_oracle_correlation_matches_reference_fixture():
    g = golden("g9_correlation.npz")
    out = O.correlation(torch.from_numpy(g["x1"]), torch.from_numpy(g["x2"]), 4)
    if same_cpu_as_golden(g):
        assert np.array_equal(out.numpy(), g["out"])
    else:
        np.testing.assert_allclose(out.numpy(), g["out"], rtol=1e-6, atol=1e-7)


def test_oracle_dpv_fuse_matches_reference_fixture():
    g = golden("g10_dpv_fuse.npz")
    logp, dm, mk = (torch.from_numpy(g[k]) for k in ("logp", "dmaps", "masks"))
    tofuse = O.gen_dpv_withmask(dm, mk, g["d_candi"], 0.3)
    fused, logf = O.dpv_fuse(logp, dm, mk, g["d_candi"], 0.3)
    if same_cpu_as_golden(g):
        assert np.array_equal(tofuse.numpy(), g["tofuse"]) and np.array_equal(fused.numpy(), g["fused"])
        assert np.array_equal(logf.numpy(), g["logfused"])
    else:
        np.testing.assert_allclose(fused.numpy(), g["fused"], rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_hip_correlation_forward():
    g = golden("g9_correlation.npz")
    dev = torch.device("cuda:0")
    out = Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1)(
        torch.from_numpy(g["x1"]).to(dev), torch.from_numpy(g["x2"]).to(dev))
    assert out.shape == (2, 81, 12, 16)
    np.testing.assert_allclose(out.cpu().numpy(), g["out"], rtol=1e-5, atol=1e-6)  # tolerance 1e-6
    # ragged size, other radius, stride2 = 2, against the oracle restatement
    x1, x2 = torch.randn(1, 19, 21, 35), torch.randn(1, 19, 21, 35)
    got = ops.correlation(x1.to(dev), x2.to(dev), pad_size=2, max_displacement=2).cpu()
    np.testing.assert_allclose(got.numpy(), O.correlation(x1, x2, 2).numpy(), rtol=1e-5, atol=1e-6)
    got = ops.correlation(x1.to(dev), x2.to(dev), pad_size=4, max_displacement=4, stride2=2).cpu()
    want = O.correlation(x1, x2, 4).reshape(1, 9, 9, 21, 35)[:, ::2, ::2].reshape(1, 25, 21, 35)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(RuntimeError, match="kernel_size 3"):   # the reference kernel reads out of bounds there
        ops.correlation(x1.to(dev), x2.to(dev), kernel_size=3)


@pytest.mark.gpu
def test_hip_dpv_fuse():
    g = golden("g10_dpv_fuse.npz")
    dev = torch.device("cuda:0")
    fused, logf = ops.dpv_fuse(torch.from_numpy(g["logp"]).to(dev), torch.from_numpy(g["dmaps"]).to(dev),
                               torch.from_numpy(g["masks"]).to(dev), g["d_candi"], var=0.3)
    np.testing.assert_allclose(fused.cpu().numpy(), g["fused"], rtol=2e-5, atol=1e-9)
    np.testing.assert_allclose(logf.cpu().numpy(), g["logfused"], rtol=1e-5, atol=2e-5)
    assert (fused.sum(1) - 1).abs().max().item() < 1e-4


def test_oracle_inverse_warp_matches_reference_fixture():
    g = golden("g11_inverse_warp.npz")
    out, valid = O.inverse_warp(torch.from_numpy(g["img"]), torch.from_numpy(g["depth"]), torch.from_numpy(g["pose44"]),
                                torch.from_numpy(g["K"]))
    # another host's BLAS rounds K^-1 @ pix and K @ pose differently (see csrc/geometry.hpp): sample positions
    # move by an ulp and the warped random image by ~1e-4
    tol = 1e-5 if same_cpu_as_golden(g) else 5e-4
    np.testing.assert_allclose(out.numpy(), g["out44"], rtol=tol, atol=tol)
    assert (valid.numpy() != g["valid44"]).mean() < 0.01


@pytest.mark.gpu
def test_hip_inverse_warp():
    from pdepth_amd.utils import inverse_warp as iw
    g = golden("g11_inverse_warp.npz")
    dev = torch.device("cuda:0")
    img, dep, K = (torch.from_numpy(g[k]).to(dev) for k in ("img", "depth", "K"))
    for pose_key, mode, okey, vkey in (("pose44", "euler", "out44", "valid44"), ("pose6", "euler", "out6e", "valid6e"),
                                       ("pose6", "quat", "out6q", "valid6q")):
        out, valid = iw.inverse_warp(img, dep, torch.from_numpy(g[pose_key]).to(dev), K, rotation_mode=mode)
        assert out.shape == (2, 3, 20, 28) and valid.dtype == torch.bool
        np.testing.assert_allclose(out.cpu().numpy(), g[okey], rtol=1e-4, atol=2e-4)
        assert (valid.cpu().numpy() != g[vkey]).mean() < 0.01  # a coordinate exactly at +-1 may flip by rounding
    with pytest.raises(NotImplementedError):
        iw.inverse_warp(img, dep, torch.from_numpy(g["pose44"]).to(dev), K, padding_mode="border")
    with pytest.raises(AssertionError, match="wrong size for depth"):
        iw.inverse_warp(img, dep[:, None], torch.from_numpy(g["pose44"]).to(dev), K)


def test_oracle_dpv_variance_properties():
    """The variance restatement (trainer/default_trainer.py:333-336) has no reference callable to pin it; check it
    against a float64 evaluation of the same formula and against the two limiting cases."""
    rng = np.random.default_rng(3)
    d = np.linspace(5.0, 40.0, 16)
    logits = torch.from_numpy(rng.normal(size=(1, 16, 5, 7)).astype(np.float32) * 3)
    logp = torch.log_softmax(logits, dim=1)
    mean, var = O.dpv_variance(logp, d)
    z = np.exp(logp[0].numpy().astype(np.float64))
    m64 = (d[:, None, None] * z).sum(0)
    v64 = (((d[:, None, None] - m64) ** 2) * z).sum(0)
    np.testing.assert_allclose(mean.numpy(), m64, rtol=1e-5)
    np.testing.assert_allclose(var.numpy(), v64, rtol=1e-4, atol=1e-4)
    onehot = torch.full((1, 16, 2, 2), -1e4)
    onehot[0, 5] = 0.0
    m1, v1 = O.dpv_variance(onehot, d)
    assert np.allclose(m1.numpy(), d[5]) and np.allclose(v1.numpy(), 0.0, atol=1e-6)


@pytest.mark.gpu
def test_hip_dpv_moments():
    dev = torch.device("cuda")
    rng = np.random.default_rng(4)
    for (B, D, H, W) in ((2, 64, 33, 47), (1, 16, 8, 8), (3, 7, 5, 130)):
        d = np.sort(rng.uniform(3.0, 60.0, size=D))
        logits = torch.from_numpy(rng.normal(size=(B, D, H, W)).astype(np.float32) * 4)
        logp = torch.log_softmax(logits, dim=1)
        mean, var = ops.dpv_moments(logp.to(dev), d, BV_log=True)
        for b in range(B):
            om, ov = O.dpv_variance(logp[b:b + 1], d)
            assert float((mean[b].cpu() - om).abs().max()) < 1e-4
            assert float(((var[b].cpu() - ov).abs() / (1.0 + ov.abs())).max()) < 1e-4
        # linear-space input gives the same moments
        m2, v2 = ops.dpv_moments(torch.exp(logp).to(dev), d, BV_log=False)
        assert float((m2 - mean).abs().max()) < 1e-4 and float(((v2 - var).abs() / (1.0 + var.abs())).max()) < 1e-4
        # the mean is dpv_to_depthmap
        assert float((ops.dpv_expect(logp.to(dev), d, BV_log=True) - mean).abs().max()) < 1e-4
    with pytest.raises(RuntimeError):
        ops.dpv_moments(torch.zeros(1, 4, 2, 2, device=dev), np.ones(5))


def test_oracle_correlation_backward_matches_reference_fixture():
    """Autograd through the oracle's restatement vs autograd through the reference's correlation_native (fixture g12)."""
    g = golden("g12_correlation_backward.npz")
    x1 = torch.from_numpy(g["x1"]).requires_grad_(True)
    x2 = torch.from_numpy(g["x2"]).requires_grad_(True)
    O.correlation(x1, x2, 4).backward(torch.from_numpy(g["grad_out"]))
    np.testing.assert_allclose(x1.grad.numpy(), g["grad_x1"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(x2.grad.numpy(), g["grad_x2"], rtol=1e-6, atol=1e-7)


@pytest.mark.gpu
def test_hip_correlation_backward():
    dev = torch.device("cuda")
    g = golden("g12_correlation_backward.npz")
    x1 = torch.from_numpy(g["x1"]).to(dev).requires_grad_(True)
    x2 = torch.from_numpy(g["x2"]).to(dev).requires_grad_(True)
    out = Correlation(pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1)(x1, x2)
    out.backward(torch.from_numpy(g["grad_out"]).to(dev))
    np.testing.assert_allclose(x1.grad.cpu().numpy(), g["grad_x1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x2.grad.cpu().numpy(), g["grad_x2"], rtol=1e-5, atol=1e-6)
    # only one input needs a gradient; a stride-2 displacement grid against the oracle's autograd
    rng = np.random.default_rng(5)
    a = torch.from_numpy(rng.normal(size=(1, 6, 9, 13)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=(1, 6, 9, 13)).astype(np.float32))
    go = torch.from_numpy(rng.normal(size=(1, 25, 9, 13)).astype(np.float32))
    ad = a.to(dev).requires_grad_(True)
    ops.correlation(ad, b.to(dev), 4, 1, 4, 1, 2).backward(go.to(dev))
    ac = a.clone().requires_grad_(True)
    full = O.correlation(ac, b, 4)  # 81 channels; stride 2 keeps displacements -4, -2, 0, 2, 4
    idx = [i * 9 + j for i in range(0, 9, 2) for j in range(0, 9, 2)]
    full[:, idx].backward(go)
    np.testing.assert_allclose(ad.grad.cpu().numpy(), ac.grad.numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
def test_hip_correlation_backward_ragged_tiles_and_channel_chunks():
    """More than one 16x16 tile in both directions, a channel count that is not a multiple of the chunk of 8, and each
    combination of requested gradients, against autograd through the oracle."""
    dev = torch.device("cuda")
    rng = np.random.default_rng(6)
    for (C, H, W, r, s2) in ((19, 21, 35, 2, 1), (8, 33, 18, 4, 2), (3, 16, 16, 1, 1)):
        nd = 2 * r // s2 + 1
        a = torch.from_numpy(rng.normal(size=(2, C, H, W)).astype(np.float32))
        b = torch.from_numpy(rng.normal(size=(2, C, H, W)).astype(np.float32))
        go = torch.from_numpy(rng.normal(size=(2, nd * nd, H, W)).astype(np.float32))
        ac, bc = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
        full = O.correlation(ac, bc, r)
        idx = [i * (2 * r + 1) + j for i in range(0, 2 * r + 1, s2) for j in range(0, 2 * r + 1, s2)]
        full[:, idx].backward(go)
        for need in ((True, True), (True, False), (False, True)):
            ad, bd = a.to(dev).requires_grad_(need[0]), b.to(dev).requires_grad_(need[1])
            ops.correlation(ad, bd, r, 1, r, 1, s2).backward(go.to(dev))
            if need[0]:
                np.testing.assert_allclose(ad.grad.cpu().numpy(), ac.grad.numpy(), rtol=1e-5, atol=2e-6)
            if need[1]:
                np.testing.assert_allclose(bd.grad.cpu().numpy(), bc.grad.numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_hip_dpv_fuse_every_depth_count_path():
    """D <= 64 and D <= 128 keep the column in registers, deeper volumes re-read it: all against the oracle."""
    dev = torch.device("cuda")
    rng = np.random.default_rng(7)
    for (B, D, H, W) in ((2, 64, 9, 70), (1, 33, 5, 300), (1, 128, 7, 40), (1, 100, 4, 9), (1, 130, 6, 11)):
        d = np.linspace(3.0, 60.0, D)
        logp = torch.log_softmax(torch.from_numpy(rng.normal(size=(B, D, H, W)).astype(np.float32) * 3), dim=1)
        masks = torch.from_numpy((rng.uniform(size=(B, 1, H, W)) > 0.6).astype(np.float32))
        dmaps = torch.from_numpy(rng.uniform(3.0, 60.0, size=(B, H, W)).astype(np.float32)) * masks[:, 0]
        want_fused, want_log = O.dpv_fuse(logp, dmaps, masks, d, 0.3)
        fused, logf = ops.dpv_fuse(logp.to(dev), dmaps.to(dev), masks.to(dev), d, var=0.3)
        np.testing.assert_allclose(fused.cpu().numpy(), want_fused.numpy(), rtol=2e-5, atol=1e-9)
        np.testing.assert_allclose(logf.cpu().numpy(), want_log.numpy(), rtol=1e-5, atol=3e-5)
        only_log = ops.dpv_fuse(logp.to(dev), dmaps.to(dev), masks.to(dev), d, var=0.3, want_fused=False)
        assert only_log[0] is None and torch.equal(only_log[1], logf)


def _g17_case(g, mode, rot, dev=None):
    from pdepth_amd.utils import inverse_warp as iw
    t = lambda k: torch.from_numpy(g[k]) if dev is None else torch.from_numpy(g[k]).to(dev)
    img, depth, pose = t("img").requires_grad_(True), t("depth").requires_grad_(True), t("pose6").requires_grad_(True)
    return iw, img, depth, pose, t("K"), t("grad_out")


def test_oracle_inverse_warp_backward_matches_reference_fixture():
    """Autograd through the oracle's restatement (+ the host's pose_vec2mat) vs autograd through the reference's
    inverse_warp (fixture g17): gradients to the image, the depth map, the 6-DoF pose and the intrinsics."""
    g = golden("g17_inverse_warp_backward.npz")
    for mode in ("bilinear", "nearest"):
        for rot in ("euler", "quat"):
            iw, img, depth, pose, K, gout = _g17_case(g, mode, rot)
            pm = iw.pose_vec2mat(pose, rot)
            out, valid = O.inverse_warp(img, depth, pm, K, mode)
            (out * gout).sum().backward()
            tag = mode + "_" + rot
            np.testing.assert_allclose(out.detach().numpy(), g[tag + "_out"], rtol=1e-4, atol=2e-4)
            assert (valid.numpy() != g[tag + "_valid"]).mean() < 0.01
            np.testing.assert_allclose(img.grad.numpy(), g[tag + "_gimg"], rtol=1e-4, atol=2e-4)
            if mode == "bilinear":
                gd, gp = g[tag + "_gdepth"], g[tag + "_gpose"]
                assert np.abs(depth.grad.numpy() - gd).max() <= 2e-3 * np.abs(gd).max()
                assert np.abs(pose.grad.numpy() - gp).max() <= 2e-3 * np.abs(gp).max()
    depth, K = torch.from_numpy(g["depth"]).requires_grad_(True), torch.from_numpy(g["K"]).requires_grad_(True)
    p44 = torch.from_numpy(g["pose44"]).requires_grad_(True)
    out, _ = O.inverse_warp(torch.from_numpy(g["img"]), depth, p44, K)
    (out * torch.from_numpy(g["grad_out"])).sum().backward()
    for got, key in ((depth.grad, "p44_gdepth"), (p44.grad, "p44_gpose"), (K.grad, "p44_gK")):
        assert np.abs(got.numpy() - g[key]).max() <= 2e-3 * np.abs(g[key]).max(), key


@pytest.mark.gpu
def test_hip_inverse_warp_backward():
    """utils.inverse_warp under autograd (the way losses/loss_blocks.py:116,151 call the reference's) against the
    gradients autograd computes through the reference (fixture g17): image (atomic scatter), depth, pose, intrinsics."""
    g = golden("g17_inverse_warp_backward.npz")
    dev = torch.device("cuda")
    for mode in ("bilinear", "nearest"):
        for rot in ("euler", "quat"):
            iw, img, depth, pose, K, gout = _g17_case(g, mode, rot, dev)
            out, valid = iw.inverse_warp(img, depth, pose, K, mode, rot)
            assert out.requires_grad and not valid.requires_grad and valid.dtype == torch.bool
            (out * gout).sum().backward()
            tag = mode + "_" + rot
            np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + "_out"], rtol=1e-4, atol=2e-4, err_msg=tag)
            assert (valid.cpu().numpy() != g[tag + "_valid"]).mean() < 0.01
            np.testing.assert_allclose(img.grad.cpu().numpy(), g[tag + "_gimg"], rtol=1e-4, atol=3e-4, err_msg=tag)
            gd, gp = g[tag + "_gdepth"], g[tag + "_gpose"]
            if mode == "bilinear":
                assert np.abs(depth.grad.cpu().numpy() - gd).max() <= 2e-3 * np.abs(gd).max(), tag
                assert np.abs(pose.grad.cpu().numpy() - gp).max() <= 2e-3 * np.abs(gp).max(), tag
            else:   # the reference's autograd yields exact zeros through mode='nearest'
                assert float(depth.grad.abs().max()) == 0.0 and float(pose.grad.abs().max()) == 0.0
    # a 4x4 pose, gradients to the intrinsics as well; the image does not require grad here
    depth = torch.from_numpy(g["depth"]).to(dev).requires_grad_(True)
    K = torch.from_numpy(g["K"]).to(dev).requires_grad_(True)
    p44 = torch.from_numpy(g["pose44"]).to(dev).requires_grad_(True)
    out, _ = iw.inverse_warp(torch.from_numpy(g["img"]).to(dev), depth, p44, K)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["p44_out"], rtol=1e-4, atol=2e-4)
    (out * torch.from_numpy(g["grad_out"]).to(dev)).sum().backward()
    for got, key in ((depth.grad, "p44_gdepth"), (p44.grad, "p44_gpose"), (K.grad, "p44_gK")):
        assert np.abs(got.cpu().numpy() - g[key]).max() <= 2e-3 * np.abs(g[key]).max(), key
    # without grad the plain kernel path is taken and gives the same forward
    with torch.no_grad():
        o2, _ = iw.inverse_warp(torch.from_numpy(g["img"]).to(dev), depth, p44, K)
    assert torch.equal(o2, out.detach())
