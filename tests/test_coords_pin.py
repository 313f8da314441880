"""CPU suite: the rounding order documented in csrc/geometry.hpp is what torch-CPU does HERE.

The HIP kernels reproduce the reference's sampling positions bit-for-bit only if the host
BLAS / ATen kernels round the way geometry.hpp assumes (fma chain in sgemm, (p1+p2)+p0 in
sgemv, fma un-normalise).  This test evaluates those formulas in numpy (fma emulated through
float64) and compares with the oracle's torch ops, so a CPU whose MKL dispatch differs shows
up here rather than as an unexplained 1e-4 depth difference in the GPU suite.
"""
import numpy as np
import pytest
import torch

import pdepth_amd
from pdepth_amd import synth
from oracle import ref_cpu as O

f32 = np.float32


def fma(a, b, c):
    return (np.asarray(a, f32).astype(np.float64) * np.asarray(b, f32).astype(np.float64)
            + np.asarray(c, f32).astype(np.float64)).astype(f32)


def dot3(separate, a0, b0, a1, b1, a2, b2):
    p0 = (np.asarray(a0, f32) * np.asarray(b0, f32)).astype(f32)
    if separate:
        p1 = (np.asarray(a1, f32) * np.asarray(b1, f32)).astype(f32)
        p2 = (np.asarray(a2, f32) * np.asarray(b2, f32)).astype(f32)
        return ((p0 + p1).astype(f32) + p2).astype(f32)
    return fma(a2, b2, fma(a1, b1, p0))


def emulate_coords(K, R, t, rays, d_candi, cx, cy, h, w):
    """csrc/geometry.hpp in numpy, with the BLAS rounding mode the host probe selected."""
    sep = pdepth_amd._native.host_blas_mode() == pdepth_amd._native.BLAS_SEPARATE
    KR = np.zeros((3, 3), f32)
    kt = np.zeros(3, f32)
    for i in range(3):
        for j in range(3):
            KR[i, j] = dot3(sep, K[i, 0], R[0, j], K[i, 1], R[1, j], K[i, 2], R[2, j])
        p = [f32(K[i, k] * t[k]) for k in range(3)]
        kt[i] = f32(f32(p[0] + p[1]) + p[2]) if sep else f32(f32(p[1] + p[2]) + p[0])
    T2 = np.stack([dot3(sep, KR[i, 0], rays[0], KR[i, 1], rays[1], KR[i, 2], rays[2]) for i in range(3)])
    d = np.asarray(d_candi).astype(f32)
    P = (kt[None, :, None] + (T2[None] * d[:, None, None]).astype(f32)).astype(f32)
    den = (P[:, 2] + f32(1e-10)).astype(f32)
    u = (P[:, 0] / den).astype(f32)
    v = (P[:, 1] / den).astype(f32)
    gx = ((u - cx).astype(f32) / cx).astype(f32)
    gy = ((v - cy).astype(f32) / cy).astype(f32)
    ix = fma((gx + f32(1)).astype(f32), f32(w / 2), f32(-0.5))
    iy = fma((gy + f32(1)).astype(f32), f32(h / 2), f32(-0.5))
    return gx, gy, ix, iy


@pytest.mark.parametrize("pose,h,w,off", [("mono", 60, 100, 1.3), ("stereo", 64, 128, 0.0), ("mono", 64, 96, -0.4)])
def test_normalised_grid_matches_pinned_rounding(pose, h, w, off):
    it = synth.make_item(11, C=1, D=16, H=h, W=w, V=1, pose=pose, cx_off=off, cy_off=-off / 2)
    K = it["K"]
    cx, cy = K.numpy()[0, 2], K.numpy()[1, 2]
    d32 = torch.from_numpy(it["d_candi"].astype(f32))
    g = O.plane_coords(K, it["R"][0], it["t"][0], it["rays"], d32, cx, cy).numpy()
    gx, gy, _, _ = emulate_coords(K.numpy(), it["R"][0].numpy(), it["t"][0].numpy(), it["rays"].numpy(),
                                  it["d_candi"], cx, cy, h, w)
    bad = int((gx != g[..., 0]).sum() + (gy != g[..., 1]).sum())
    assert bad == 0, f"{bad} of {2 * gx.size} grid coordinates round differently on this CPU"


def test_bilinear_sampler_matches_pinned_rounding():
    """grid_sample == fma(se_v,se, fma(sw_v,sw, fma(ne_v,ne, nw_v*nw))) at ix = fma(g+1, w/2, -.5)."""
    import torch.nn.functional as F
    h, w, D = 60, 100, 8
    it = synth.make_item(12, C=1, D=D, H=h, W=w, V=1, pose="mono", cx_off=0.9, cy_off=-0.3)
    K = it["K"]
    cx, cy = K.numpy()[0, 2], K.numpy()[1, 2]
    d32 = torch.from_numpy(it["d_candi"].astype(f32))
    g = O.plane_coords(K, it["R"][0], it["t"][0], it["rays"], d32, cx, cy)
    img = it["src"][0]  # [1,h,w]
    out = F.grid_sample(img[None].repeat(D, 1, 1, 1), g.reshape(D, h, w, 2), mode="bilinear",
                        padding_mode="zeros", align_corners=False).numpy()[:, 0]
    gx, gy = g[..., 0].numpy(), g[..., 1].numpy()
    ix = fma((gx + f32(1)).astype(f32), f32(w / 2), f32(-0.5))
    iy = fma((gy + f32(1)).astype(f32), f32(h / 2), f32(-0.5))
    x0, y0 = np.floor(ix), np.floor(iy)
    wx = (ix - x0).astype(f32); ex = (f32(1) - wx).astype(f32)
    ny = (iy - y0).astype(f32); sy = (f32(1) - ny).astype(f32)
    xi, yi = x0.astype(np.int64).reshape(D, h, w), y0.astype(np.int64).reshape(D, h, w)
    im = img.numpy()[0]

    def tap(xx, yy):
        m = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
        return np.where(m, im[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)], f32(0))

    nw, ne, sw, se = [z.reshape(D, h, w) for z in ((sy * ex).astype(f32), (sy * wx).astype(f32),
                                                   (ny * ex).astype(f32), (ny * wx).astype(f32))]
    r = fma(tap(xi + 1, yi + 1), se, fma(tap(xi, yi + 1), sw, fma(tap(xi + 1, yi), ne, (tap(xi, yi) * nw).astype(f32))))
    bad = int((r != out).sum())
    assert bad == 0, f"{bad} of {out.size} samples round differently on this CPU"
