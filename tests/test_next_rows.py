"""DPV Bayesian fusion (rank 2) and the correlation op forward (rank 3)."""
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
    with pytest.raises(RuntimeError, match="unsupported configuration"):
        ops.correlation(x1.to(dev), x2.to(dev), kernel_size=3)
    with pytest.raises(RuntimeError, match="backward is not implemented"):
        Correlation()(x1.to(dev).requires_grad_(), x2.to(dev))


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
