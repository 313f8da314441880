"""Randomized shapes for the DPV / warp / correlation kernels (every vector-width, plane-count and raggedness path)
against plain torch formulas and the oracle -- the counterpart of the sweep fuzz tests for the other entry points."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import pdepth_amd  # noqa: F401
from pdepth_amd import ops, synth
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")


def _shapes(rng, n, dmax=140):
    for i in range(n):
        B = int(rng.integers(1, 4))
        D = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65, 100, 128, 129, dmax])) if i % 3 else int(rng.integers(1, dmax + 1))
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        if i % 4 == 0:
            W = 4 * int(rng.integers(1, 20))          # the 16-byte paths
        if i % 8 == 0:
            H = 4 * int(rng.integers(1, 10))
        yield B, D, H, W


def test_dpv_reduce_family_on_random_shapes():
    rng = np.random.default_rng(2024)
    gen = torch.Generator().manual_seed(2024)
    for B, D, H, W in _shapes(rng, 120):
        x = torch.randn(B, D, H, W, generator=gen) * float(rng.choice([0.1, 1.0, 5.0, 30.0]))
        add = torch.randn(B, D, H, W, generator=gen) if rng.integers(0, 2) else None
        dc = np.sort(rng.uniform(0.5, 60.0, size=D)) if rng.integers(0, 2) else rng.uniform(0.5, 60.0, size=D)
        dt = torch.tensor(dc, dtype=torch.float32)[None, :, None, None]
        z = x + (add if add is not None else 0.0)
        want_lp = F.log_softmax(z, dim=1)
        p64 = torch.softmax(z.double(), dim=1)
        mean64 = (dt.double() * p64).sum(1)
        var64 = (((dt.double() - mean64[:, None]) ** 2) * p64).sum(1)
        xd, ad = x.to(DEV), (add.to(DEV) if add is not None else None)
        quarter = H >= 4 and W >= 4
        r = ops.dpv_reduce_ex(xd, dc, addend=ad, want_logp=True, want_prob=True, want_depth=True, want_var=True, want_quarter=quarter)
        tag = f"B={B} D={D} {H}x{W} addend={add is not None}"
        assert (r["logp"].cpu() - want_lp).abs().max().item() < 3e-5 * max(1.0, float(z.abs().max()) / 10), tag
        assert (r["prob"].cpu().double() - p64).abs().max().item() < 2e-6, tag
        assert (r["depth"].cpu().double() - mean64).abs().max().item() < 1e-4, tag
        assert ((r["var"].cpu().double() - var64).abs() / (1.0 + var64)).max().item() < 2e-4, tag
        if quarter:
            assert torch.equal(r["quarter"], F.interpolate(r["logp"], scale_factor=0.25, mode="nearest")), tag
        if add is None:
            lp, dp = ops.dpv_reduce(xd, dc)
            assert (lp - r["logp"]).abs().max().item() < 2e-6 + 4e-7 * float(z.abs().max()) and (dp - r["depth"]).abs().max().item() < 5e-5, tag   # (two kernels, two summation layouts)
            lp2, _ = ops.dpv_reduce(xd.clone(), dc, inplace=True)
            assert torch.equal(lp2, lp), tag
            for bv_log, vol in ((True, r["logp"]), (False, r["prob"])):
                e = ops.dpv_expect(vol, dc, BV_log=bv_log)
                assert (e.cpu().double() - mean64).abs().max().item() < 1e-4, tag
                m, v = ops.dpv_moments(vol, dc, BV_log=bv_log)
                assert (m - e).abs().max().item() < 5e-5, tag
                assert ((v.cpu().double() - var64).abs() / (1.0 + var64)).max().item() < 2e-4, tag


def test_dpv_fuse_and_ufield_on_random_shapes():
    rng = np.random.default_rng(7)
    gen = torch.Generator().manual_seed(7)
    for B, D, H, W in _shapes(rng, 40, dmax=140):
        D = max(D, 8)   # (candidates at most 7 m apart: a sparse depth 7.9 m from every candidate sits on the last fp32
        dc = np.linspace(2.0, 50.0, D)   #  denormal of the reference's Gaussian, where 0 / 0 and 1e-45 / 1e-45 decide the result)
        logp = torch.log_softmax(torch.randn(B, D, H, W, generator=gen) * 3, dim=1)
        masks = (torch.rand(B, 1, H, W, generator=gen) > 0.5).float()
        dmaps = (torch.rand(B, H, W, generator=gen) * 45 + 3) * masks[:, 0]
        wf, wl = O.dpv_fuse(logp, dmaps, masks, dc, 0.3)
        gf, gl = ops.dpv_fuse(logp.to(DEV), dmaps.to(DEV), masks.to(DEV), dc, var=0.3)
        tag = f"B={B} D={D} {H}x{W}"
        np.testing.assert_allclose(gf.cpu().numpy(), wf.numpy(), rtol=3e-5, atol=1e-9, err_msg=tag)
        np.testing.assert_allclose(gl.cpu().numpy(), wl.numpy(), rtol=1e-5, atol=4e-5, err_msg=tag)
    # uncertainty field: shapes the fixture does not have (two columns, two rows, W not a multiple of 64, D not of 4).
    # The masks are thresholds on a depth map that the kernel sums in another order than the oracle, so a pixel within
    # 1e-5 of a threshold may flip its column: at most a few columns in all may differ.
    from pdepth_amd.utils import img_utils
    cols = bad_cols = 0
    for (B, D, H, W) in ((1, 5, 9, 2), (2, 7, 2, 13), (1, 33, 30, 65), (3, 64, 17, 130), (1, 130, 12, 40)):
        dc = np.linspace(3.0, 40.0, D)
        logp = torch.log_softmax(torch.randn(B, D, H, W, generator=gen) * 2, dim=1)
        intr = torch.tensor([[0.9 * W, 0.0, W / 2.0], [0.0, 0.9 * W, H / 2.0], [0.0, 0.0, 1.0]])
        for ang in (0, 2):
            for b in range(B):
                cfgx = {"unc_ang": ang, "unc_shift": -5.0, "unc_span": 10.0}
                plane, dz = img_utils.gen_ufield(logp[b:b + 1].to(DEV), dc, intr.to(DEV), BV_log=True, cfgx=cfgx)
                wp, wd = O.gen_ufield(logp[b:b + 1], dc, intr, ang, -5.0, 10.0, BV_log=True)
                got, want = plane.cpu().numpy().reshape(D, W), wp.numpy().reshape(D, W)
                same = np.isclose(got, want, rtol=3e-5, atol=1e-6, equal_nan=True).all(axis=0)
                cols += W
                bad_cols += int((~same).sum())
    assert cols > 400 and bad_cols <= 0.02 * cols, (bad_cols, cols)


def test_correlation_and_warps_on_random_shapes():
    rng = np.random.default_rng(11)
    gen = torch.Generator().manual_seed(11)
    for i in range(30):
        B, C, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 40)), int(rng.integers(1, 50)), int(rng.integers(1, 70))
        r, s2 = [(1, 1), (2, 1), (3, 1), (4, 1), (2, 2), (4, 2), (4, 4)][i % 7]
        x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
        md = r
        nd = 2 * (md // s2) + 1
        full = O.correlation(x1, x2, md)
        idx = [a * (2 * md + 1) + c for a in range(0, 2 * md + 1, s2) for c in range(0, 2 * md + 1, s2)]
        got = ops.correlation(x1.to(DEV), x2.to(DEV), pad_size=md, max_displacement=md, stride2=s2).cpu()
        assert got.shape == (B, nd * nd, H, W)
        np.testing.assert_allclose(got.numpy(), full[:, idx].numpy(), rtol=1e-5, atol=2e-6, err_msg=str((B, C, H, W, r, s2)))
    from pdepth_amd.utils import inverse_warp as iw
    for i in range(20):
        B, C, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 6)), int(rng.integers(2, 40)), int(rng.integers(2, 60))
        img = torch.randn(B, C, H, W, generator=gen)
        dep = torch.rand(B, H, W, generator=gen) * 30 + 2
        K = torch.tensor([[0.8 * W, 0.0, W / 2.0 + 0.3], [0.0, 0.8 * W, H / 2.0 - 0.2], [0.0, 0.0, 1.0]]).repeat(B, 1, 1)
        pose = torch.eye(4).repeat(B, 1, 1)
        pose[:, :3, 3] = torch.from_numpy(rng.uniform(-1.0, 1.0, size=(B, 3)).astype(np.float32))
        for mode in ("bilinear", "nearest"):
            wo, wv = O.inverse_warp(img, dep, pose, K, mode)
            go, gv = iw.inverse_warp(img.to(DEV), dep.to(DEV), pose.to(DEV), K.to(DEV), mode)
            near_edge = 0.02 if mode == "nearest" else 0.0   # a sample within rounding of a texel boundary may pick the other texel
            diff = (go.cpu() - wo).abs()
            assert float((diff > 2e-4).float().mean()) <= near_edge, (mode, B, C, H, W, float(diff.max()))
            assert (gv.cpu() != wv).float().mean().item() < 0.02


def test_correlation_forward_matrix_pipe_on_random_shapes():
    """The matrix-pipe forward of the fast configuration (kernel 1, stride1 1, pad = max_displacement; extras.hip) over channel
    counts on both sides of its three instantiations (C <= 64 / 128 / 256), image widths around the 16-pixel groups and the
    64-pixel workgroups, every radius and displacement stride it takes -- against the restatement of the reference's kernel."""
    rng = np.random.default_rng(77)
    gen = torch.Generator().manual_seed(77)
    for i in range(60):
        C = int(rng.choice([1, 3, 4, 5, 31, 64, 65, 67, 128, 129, 200, 256])) if i % 2 else int(rng.integers(1, 257))
        H = int(rng.integers(1, 24))
        W = int(rng.choice([1, 7, 15, 16, 17, 31, 33, 63, 64, 65, 80, 100, 129]))
        r = int(rng.integers(1, 5))
        s2 = int(rng.choice([1, 2, 3, 4]))
        if r * s2 > 16:
            s2 = 1
        md = r * s2
        B = int(rng.integers(1, 3))
        x1, x2 = torch.randn(B, C, H, W, generator=gen), torch.randn(B, C, H, W, generator=gen)
        want = O.correlation_general(x1, x2, md, 1, md, 1, s2)
        got = ops.correlation(x1.to(DEV), x2.to(DEV), pad_size=md, kernel_size=1, max_displacement=md, stride1=1, stride2=s2).cpu()
        assert got.shape == want.shape, (C, H, W, r, s2)
        err = float((got - want).abs().max())
        assert err < 2e-6, (C, H, W, r, s2, err)
