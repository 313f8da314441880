"""CPU oracle for the plane-sweep / DPV hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch CPU restatement (plain torch fp32 ops, CPU tensors)
of the reference algorithm.  It exists so that the HIP kernels can be checked
against the reference's PyTorch-CPU numerics on a machine where the reference
itself is absent (the GPU box).  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it; the product package
(``probabilistic-depth_amd``) never does, and has no CPU fallback.

Parity pin: ``tests/golden/make_golden.py`` imports the real reference from
``/root/reference`` (build container only) and stores its outputs; the
``-m "not gpu"`` suite asserts this restatement reproduces those fixtures
bit-for-bit (same ATen ops in the same order => identical rounding).

Every function cites the reference lines it restates (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "powerf", "unit_rays", "intrinsics_from_fov", "plane_coords", "sweep_cost",
    "warp_feature", "log_dpv", "dpv_to_depthmap", "sweep_dpv", "sample_coords", "gen_dpv_withmask", "dpv_fuse",
    "correlation", "correlation_general", "inverse_warp", "dpv_variance", "gen_ufield", "sweep_dpv_exact64",
]


# --------------------------------------------------------------------------------------
# host-side producers
# --------------------------------------------------------------------------------------
def powerf(d_min, d_max, n_depth, power):
    """Depth candidates, float64.  utils/img_utils.py:80-85."""
    q = np.power(np.linspace(start=0, stop=1, num=n_depth), power)
    return np.array([d_min + (d_max - d_min) * v for v in q])


def unit_rays(width, height, hfov, vfov):
    """z=1 rays, [3, h*w] fp32, column index y*w+x.

    warping/view.py:16-30 (per pixel formula), :32-62 (table),
    kittiloader/kitti.py:311-312 (transpose/reshape/cast).
    The reference fills the table with a Python double loop in float64; the
    expression below evaluates the identical float64 formula per element.
    """
    th = math.tan(math.radians(hfov / 2.0))
    tv = math.tan(math.radians(vfov / 2.0))
    rays = np.zeros((height, width, 3))
    for y in range(height):
        yv = tv * ((2.0 * ((y + 0.5) / height)) - 1.0)
        for x in range(width):
            rays[y, x, 0] = th * ((2.0 * ((x + 0.5) / width)) - 1.0)
            rays[y, x, 1] = yv
            rays[y, x, 2] = 1.0
    flat = np.reshape(np.transpose(rays, axes=[2, 0, 1]), [3, -1])
    return torch.from_numpy(flat.astype(np.float32))


def intrinsics_from_fov(width, height, hfov, vfov):
    """KITTI-branch intrinsics at the sweep resolution, float64 3x3.

    kittiloader/kitti.py:284-293.
    """
    K = np.zeros((3, 3))
    K[2, 2] = 1.0
    K[0, 0] = (width / 2.0) / math.tan(math.radians(hfov / 2.0))
    K[0, 2] = width / 2.0
    K[1, 1] = (height / 2.0) / math.tan(math.radians(vfov / 2.0))
    K[1, 2] = height / 2.0
    return K


# --------------------------------------------------------------------------------------
# sweep geometry
# --------------------------------------------------------------------------------------
def plane_coords(K, R_v, t_v, rays, d_candi_f32, cx, cy):
    """Normalised sampling grid [D, hw, 2] for one source view.

    warping/homography.py:119-121 (term1/term2, note (K@R)@rays association),
    :185-196 (P = term1 + term2*d; P/(Pz+1e-10); (u-cx)/cx, (v-cy)/cy).
    """
    term1 = K.matmul(t_v).reshape(3, 1)
    term2 = K.matmul(R_v).matmul(rays)
    n_d = d_candi_f32.shape[0]
    P = term1.unsqueeze(0) + term2.repeat(n_d, 1, 1) * d_candi_f32.reshape(n_d, 1, 1)
    P = P / (P[:, 2, :].unsqueeze(1) + 1e-10)
    gx = (P[:, 0, :] - cx) / cx
    gy = (P[:, 1, :] - cy) / cy
    return torch.stack((gx, gy), dim=-1)


def sample_coords(K, R_v, t_v, rays, d_candi, cx, cy, h, w):
    """Un-normalised pixel coordinates (ix, iy) [D, hw] that grid_sample uses.

    ATen CPU vectorised grid_sampler_2d, align_corners=False
    (ATen/native/cpu/GridSamplerKernel.cpp ComputeLocation::unnormalize):
    ix = (gx + 1) * (w / 2) - 0.5, which the build contracts into ONE fma (pinned by
    tests/test_coords_pin.py against F.grid_sample itself).  The fma is evaluated here
    through float64 (exact product, one rounding of the sum, then the cast).
    Used by tests to check a kernel's sample positions; not part of the reference's surface.
    """
    d32 = torch.from_numpy(np.asarray(d_candi).astype(np.float32))
    g = plane_coords(K, R_v, t_v, rays, d32, cx, cy)
    ix = ((g[..., 0] + 1).double() * (w / 2) - 0.5).float()
    iy = ((g[..., 1] + 1).double() * (h / 2) - 0.5).float()
    return ix, iy


def _warp_all_planes(src_view, d_candi_f32, K, R_v, t_v, rays, cx, cy, h, w):
    """Bilinear warp of one source view into every depth plane -> [D, C, h, w].

    warping/homography.py:123 (repeat over D), :170-198 (_back_warp_homo_parallel;
    grid_sample bilinear / zeros / default align_corners=False).
    """
    n_d = d_candi_f32.shape[0]
    grid = plane_coords(K, R_v, t_v, rays, d_candi_f32, cx, cy).reshape(n_d, h, w, 2)
    stacked = src_view.repeat(n_d, 1, 1, 1)
    return F.grid_sample(stacked, grid, mode="bilinear", padding_mode="zeros", align_corners=False)


def sweep_cost(feat_ref, feat_src, d_candi, R, t, K, rays, cx, cy, sigma, metric="L2"):
    """Plane-sweep cost volume of ONE batch item -> [1, D, h, w].

    warping/homography.py:98-135 (est_swp_volume_v4), :80-86 (L2 / L1 distance).
    feat_ref [1,C,h,w]; feat_src [1,V,C,h,w]; R [V,3,3]; t [V,3]; K [3,3] fp32 tensor;
    rays [3,hw]; cx, cy = np.float32 principal point read from the numpy copy of K
    (models/models.py:538).
    """
    h, w = feat_ref.shape[2], feat_ref.shape[3]
    d32 = torch.from_numpy(np.asarray(d_candi).astype(np.float32))
    cost = torch.zeros(1, d32.shape[0], h, w)
    for v in range(feat_src.shape[1]):
        warped = _warp_all_planes(feat_src[:, v], d32, K, R[v], t[v], rays, cx, cy, h, w)
        if metric == "L2":
            dist = torch.sum((warped - feat_ref) ** 2, 1)
        elif metric == "L1":
            dist = torch.sum(torch.abs(warped - feat_ref), 1)
        else:
            raise Exception("undefined metric for feature distance ...")
        cost[0] = cost[0] + dist / sigma
    return cost


def sweep_cost_at(feat_ref, feat_src, d_candi, R, t, K, rays, cx, cy, sigma, idx, metric="L2"):
    """sweep_cost() for the pixels `idx` (flat indices, LongTensor [n]) only -> [1, D, n]: the same ops on the same values
    -- (K@R)@rays is formed for every pixel and THEN restricted (a matmul over fewer columns may take another BLAS path),
    everything after it is elementwise per pixel; grid_sample is called plane by plane on the one source view instead of on
    its D-fold repeat (18 GB per view at BASELINE config 5).  tests/test_oracle_subset.py pins it to sweep_cost()."""
    h, w = feat_ref.shape[2], feat_ref.shape[3]
    d32 = torch.from_numpy(np.asarray(d_candi).astype(np.float32))
    n_d, n = d32.shape[0], idx.shape[0]
    ref_at = feat_ref.reshape(1, feat_ref.shape[1], h * w)[:, :, idx].reshape(1, -1, 1, n)
    cost = torch.zeros(1, n_d, n)
    for v in range(feat_src.shape[1]):
        term1 = K.matmul(t[v]).reshape(3, 1)
        term2 = K.matmul(R[v]).matmul(rays)[:, idx]
        P = term1.unsqueeze(0) + term2.repeat(n_d, 1, 1) * d32.reshape(n_d, 1, 1)
        P = P / (P[:, 2, :].unsqueeze(1) + 1e-10)
        grid = torch.stack(((P[:, 0, :] - cx) / cx, (P[:, 1, :] - cy) / cy), dim=-1).reshape(n_d, 1, n, 2)
        for k in range(n_d):
            warped = F.grid_sample(feat_src[:, v], grid[k:k + 1], mode="bilinear", padding_mode="zeros", align_corners=False)
            if metric == "L2":
                dist = torch.sum((warped - ref_at) ** 2, 1)
            elif metric == "L1":
                dist = torch.sum(torch.abs(warped - ref_at), 1)
            else:
                raise Exception("undefined metric for feature distance ...")
            cost[0, k] = cost[0, k] + dist.reshape(n) / sigma
    return cost


def warp_feature(feat_src, d_candi, R, t, K, rays, cx, cy):
    """Channel i of every view warped with depth plane i -> [1, V, D, h, w].

    warping/homography.py:137-168: the reference warps all D x C planes and keeps
    the diagonal [i, i]; restated the same (wasteful) way so rounding is identical.
    """
    if feat_src.shape[0] != 1:
        raise Exception("Warped Accum Error")
    h, w = feat_src.shape[3], feat_src.shape[4]
    d32 = torch.from_numpy(np.asarray(d_candi).astype(np.float32))
    out = torch.zeros(feat_src.shape)
    for v in range(feat_src.shape[1]):
        warped = _warp_all_planes(feat_src[:, v], d32, K, R[v], t[v], rays, cx, cy, h, w)
        idx = torch.arange(d32.shape[0])
        out[0, v] = warped[idx, idx]
    return out


# --------------------------------------------------------------------------------------
# DPV reduction
# --------------------------------------------------------------------------------------
def log_dpv(logits):
    """log-softmax over the depth axis.  models/models.py:560,637,694; packnet.py:394."""
    return F.log_softmax(logits, dim=1)


def dpv_to_depthmap(dpv, d_candi, BV_log=False):
    """E[d] over the depth axis of a [1,D,H,W] DPV -> [1,H,W].  utils/img_utils.py:52-61."""
    if dpv.shape[0] != 1:
        raise Exception("Unable to handle this case")
    z = dpv.squeeze(0)
    if BV_log:
        z = torch.exp(z)
    d = torch.tensor(d_candi).unsqueeze(1).unsqueeze(1).float()
    return torch.sum(d * z, dim=0).unsqueeze(0)


def dpv_variance(dpv_log, d_candi):
    """Mean and variance of the depth distribution of a [1,D,H,W] log-DPV -> ([H,W], [H,W]), float64.

    trainer/default_trainer.py:333-336 (evaluation loop, inline): z = exp(logDPV.squeeze(0)),
    mean = sum(d * z, 0), variance = sum((d - mean)**2 * z, 0) with d = torch.tensor(d_candi) -- a FLOAT64 tensor
    (d_candi is a float64 numpy array), so both sums are promoted to float64.  Pinned by fixture g13, which executes
    those four lines of the reference file itself (tests/golden/make_golden_r2.py)."""
    z = torch.exp(dpv_log.squeeze(0))
    d = torch.tensor(np.asarray(d_candi)).unsqueeze(1).unsqueeze(1)
    mean = torch.sum(d * z, dim=0)
    variance = torch.sum(((d - mean) ** 2) * z, dim=0)
    return mean, variance


def _convert_flowfield(flow):
    """Pixel offsets -> the normalised grid the reference hands to grid_sample.  utils/img_utils.py:170-176."""
    yv, xv = torch.meshgrid([torch.arange(0, flow.shape[1]).float(), torch.arange(0, flow.shape[2]).float()], indexing="ij")
    ystep = 2. / float(flow.shape[1] - 1)
    xstep = 2. / float(flow.shape[2] - 1)
    flow[0, :, :, 0] = -1 + xv * xstep - flow[0, :, :, 0] * xstep
    flow[0, :, :, 1] = -1 + yv * ystep - flow[0, :, :, 1] * ystep
    return flow


def _depth_to_pts(depthf, intr):
    """[1,H,W] depth -> [3,H,W] points with integer pixel coordinates.  utils/img_utils.py:111-134."""
    depth = depthf[0]
    fx, cx, fy, cy = intr[0, 0], intr[0, 2], intr[1, 1], intr[1, 2]
    yf, xf = torch.meshgrid([torch.arange(0, depth.shape[0]).float(), torch.arange(0, depth.shape[1]).float()], indexing="ij")
    yf = (yf - cy) / fy
    xf = (xf - cx) / fx
    return torch.cat([torch.mul(xf, depth).unsqueeze(0), torch.mul(yf, depth).unsqueeze(0), depth.unsqueeze(0)], 0)


def gen_ufield(dpv, d_candi, intr, unc_ang, unc_shift, unc_span, BV_log=True, mask=None, normalize=False, mind=3., quash_limit=True):
    """Uncertainty-field collapse of a [1,D,H,W] (log-)DPV -> (plane [1,D,W], masked depth map [1,H,W]).

    utils/img_utils.py:268-358 with the cfgx branch (:269-275: pshift = unc_ang rows, z band [unc_shift, unc_shift +
    unc_span], maxd 100, mind 3, quash on): shift the volume by pshift rows (grid_sample nearest over a flow field
    built with the (size-1) convention, :292-300 -- the grid is NOT the identity of align_corners=False sampling: the
    last column / rows can fall out of the image), depth maps of the shifted and the original volume (:306-307),
    height-band + range mask on the shifted points (:311-315), optional validity mask (:316-321), quash to the
    nearest surface per column (:324-331), shift the mask back (:334-338), masked depth (:338), masked column sums of
    the probabilities divided by the column's mask count (:341-349), optional min/max normalisation (:353-355).
    `mind` / `quash_limit`: the dataset branches (:277-290) -- kitti: 5 rows, band [0.6, 0.9], mind 0, no quash; ilim: no
    shift, band [1.0, 1.3], mind 3, quash.
    """
    zstart, zend, maxd = unc_shift, unc_shift + unc_span, 100.
    H, W = dpv.shape[2], dpv.shape[3]
    if unc_ang != 0:
        flow = torch.zeros((1, H, W, 2)).float()
        flow_inv = torch.zeros((1, H, W, 2)).float()
        flow[:, :, :, 1] = unc_ang
        flow_inv[:, :, :, 1] = -unc_ang
        _convert_flowfield(flow)
        _convert_flowfield(flow_inv)
        dpv_shifted = F.grid_sample(dpv, flow, mode='nearest', align_corners=False)
    else:
        dpv_shifted = dpv.clone()
    depth_shifted = dpv_to_depthmap(dpv_shifted, d_candi, BV_log=BV_log)
    depth_pred = dpv_to_depthmap(dpv, d_candi, BV_log=BV_log)
    pts = _depth_to_pts(depth_shifted, intr)
    zero_mask = (~((pts[1] > zend) | (pts[1] < zstart) | (pts[2] > maxd - 1) | (pts[2] < mind))).float()
    if mask is not None:
        if unc_ang != 0:
            mask_shifted = F.grid_sample(mask.unsqueeze(1), flow, mode='nearest', align_corners=False).squeeze(1)
        else:
            mask_shifted = mask.clone()
        zero_mask = zero_mask * mask_shifted.squeeze(0)
    if quash_limit:
        cleaned = (depth_shifted * zero_mask).squeeze(0)
        cleaned[cleaned == 0] = 1000
        min_col, _ = torch.min(cleaned, axis=0)
        quash = ((cleaned > min_col - 1.) & (cleaned < min_col + 1.)).float()
        zero_mask = zero_mask * quash
    if unc_ang != 0:
        zm_pred = F.grid_sample(zero_mask.unsqueeze(0).unsqueeze(0), flow_inv, mode='nearest', align_corners=False).squeeze(0).squeeze(0)
    else:
        zm_pred = zero_mask.clone()
    depth_zero = depth_pred * zm_pred
    zm_rep = zm_pred.repeat([len(d_candi), 1, 1]).unsqueeze(0)
    plane = torch.sum((torch.exp(dpv) if BV_log else dpv) * zm_rep, axis=2)
    plane = plane / torch.sum(zero_mask, axis=0)
    if normalize:
        minval, _ = plane.min(1)
        maxval, _ = plane.max(1)
        plane = (plane - minval) / (maxval - minval)
    return plane, depth_zero


UFIELD_DATASETS = {"kitti": dict(unc_ang=5, unc_shift=0.6, unc_span=0.3, mind=0., quash_limit=False),
                   "ilim": dict(unc_ang=0, unc_shift=1.0, unc_span=0.3, mind=3., quash_limit=True)}


def compute_unc_field(dpv_predicted, dpv_truth, d_candi, intr, mask, dataset):
    """utils/img_utils.py:178-181: the field of the ground-truth DPV (probabilities, validity mask) and of the predicted
    log-DPV (no mask), both through the dataset branch of gen_ufield; intr [1,3,3]."""
    kw = UFIELD_DATASETS[dataset]
    truth, _ = gen_ufield(dpv_truth, d_candi, intr.squeeze(0), BV_log=False, mask=mask, **kw)
    pred, debugmap = gen_ufield(dpv_predicted, d_candi, intr.squeeze(0), BV_log=True, **kw)
    return truth, pred, debugmap


def compute_unc_rmse(field_truth, field_pred, d_candi):
    """utils/img_utils.py:183-202: E[d] per column of the two [1,D,W] fields, the predicted one zeroed in the first and
    last column (:187-188), columns where either is NaN dropped (:189-191).  What the function returns is the MEAN
    ABSOLUTE difference over the remaining columns: its RMSE of :192 is overwritten by :193."""
    dt = dpv_to_depthmap(field_truth.unsqueeze(2), d_candi, BV_log=False).squeeze(0).squeeze(0)
    dp = dpv_to_depthmap(field_pred.unsqueeze(2), d_candi, BV_log=False).squeeze(0).squeeze(0).clone()
    dp[0] = 0
    dp[-1] = 0
    ok = ~torch.isnan(dt) & ~torch.isnan(dp)
    dt = torch.where(ok, dt, torch.zeros_like(dt))
    dp = torch.where(ok, dp, torch.zeros_like(dp))
    return torch.sum(torch.abs(dt * ok - dp * ok)) / torch.sum(ok)


def sweep_dpv(feat_ref, feat_src, d_candi, R, t, K, rays, cx, cy, sigma, metric="L2"):
    """cost -> log-DPV -> depth for one item (the PackNet-style fusable chain).

    models/packnet.py:380-394 (log_softmax straight on the cost volume) followed by
    trainer/default_trainer.py:232 (dpv_to_depthmap(..., BV_log=True)).
    Returns (cost [1,D,h,w], logp [1,D,h,w], depth [1,h,w]).
    """
    cost = sweep_cost(feat_ref, feat_src, d_candi, R, t, K, rays, cx, cy, sigma, metric)
    logp = log_dpv(cost)
    return cost, logp, dpv_to_depthmap(logp, d_candi, BV_log=True)


def sweep_dpv_exact64(feat_ref, feat_src, d_candi, R, t, K, rays, cx, cy, sigma, plane_chunk=8):
    """The chain of sweep_dpv() evaluated in float64 AT THE float32 SAMPLE POSITIONS of the reference (sample_coords(): the
    positions are data here, every implementation has them bit for bit): bilinear taps with zeros outside the image
    (warping/homography.py:197), squared differences summed over the channels (:80-82), / sigma summed over the views (:129),
    log_softmax over the planes (models/packnet.py:394), expectation (utils/img_utils.py:52-61).

    NOT the reference's numerics -- the yardstick for them: the distance of the float32 oracle from this volume is the
    rounding noise of the reference itself, which an implementation can only reproduce by copying its summation order
    (csrc/sweep_direct.hip does; the fast kernels are held to "no noisier than the reference", tests/util.py).
    Returns (cost [1,D,h,w], depth [1,h,w], kappa [1,h,w]) float64; kappa = sum_k p_k |d_k - E|: the change of the expected
    depth per unit of (worst-sign) error of the costs of a pixel, to first order."""
    ref = feat_ref[0].double()
    C, h, w = ref.shape
    V = feat_src.shape[1]
    d32 = np.asarray(d_candi).astype(np.float32)
    D = len(d32)
    refv = ref.reshape(1, C, h * w)
    cost = torch.zeros(D, h * w, dtype=torch.float64)
    for v in range(V):
        ixa, iya = sample_coords(K, R[v], t[v], rays, d32, cx, cy, h, w)   # [D, hw] float32
        sv = feat_src[0, v].double().reshape(C, h * w)
        for k0 in range(0, D, plane_chunk):
            ix, iy = ixa[k0:k0 + plane_chunk].double(), iya[k0:k0 + plane_chunk].double()
            n = ix.shape[0]
            valid = torch.isfinite(ix) & torch.isfinite(iy)
            ix, iy = torch.where(valid, ix, torch.zeros_like(ix)), torch.where(valid, iy, torch.zeros_like(iy))
            x0, y0 = torch.floor(ix), torch.floor(iy)
            fx, fy = ix - x0, iy - y0
            val = torch.zeros(n, C, h * w, dtype=torch.float64)
            for dy, dx, wt in ((0, 0, (1 - fx) * (1 - fy)), (0, 1, fx * (1 - fy)), (1, 0, (1 - fx) * fy), (1, 1, fx * fy)):
                xx, yy = x0 + dx, y0 + dy
                ok = (xx >= 0) & (xx <= w - 1) & (yy >= 0) & (yy <= h - 1)
                idx = torch.where(ok, yy * w + xx, torch.zeros_like(xx)).long()
                tap = sv[:, idx.reshape(-1)].reshape(C, n, h * w).permute(1, 0, 2)
                val += tap * (wt * ok)[:, None, :]
            dist = ((val - refv) ** 2).sum(1)
            dist[~valid] = float("nan")
            cost[k0:k0 + n] += dist / float(sigma)
    logp = torch.log_softmax(cost, 0)
    p = logp.exp()
    dk = torch.from_numpy(d32.astype(np.float64))[:, None]
    depth = (dk * p).sum(0)
    kappa = (p * (dk - depth[None]).abs()).sum(0)
    return cost.reshape(1, D, h, w), depth.reshape(1, h, w), kappa.reshape(1, h, w)


# --------------------------------------------------------------------------------------
# "next" rows: DPV Bayesian fusion and the correlation op
# --------------------------------------------------------------------------------------
EPSILON = torch.finfo(float).eps  # utils/img_utils.py:12


def gen_dpv_withmask(dmaps, masks, d_candi, var=0.3):
    """Gaussian soft label blended with the uniform DPV by the validity mask -> [B,D,H,W].

    utils/img_utils.py:360-375 (per item: gen_soft_label_torch :31-47 with zero_invalid=True,
    gen_uniform :49-50, blend :371, clamp :374).
    """
    out = []
    truth_var = torch.tensor(var)
    d = torch.tensor(d_candi).float()
    for b in range(dmaps.shape[0]):
        dmap = dmaps[b]
        mask = masks[b, 0].unsqueeze(0)
        dexp = d.unsqueeze(-1).unsqueeze(-1).repeat(1, dmap.shape[0], dmap.shape[1])
        sigma = torch.sqrt(truth_var)
        dists = torch.exp(-torch.pow(torch.abs(dexp - dmap), 2.0) / (2 * torch.pow(sigma, 2.0)))
        dists = dists / torch.sum(dists, dim=0)
        dists[dists != dists] = -1
        uni = torch.ones((d.shape[0], dmap.shape[0], dmap.shape[1])) / d.shape[0]
        out.append((dists * mask + uni * (1.0 - mask)).unsqueeze(0))
    return torch.clamp(torch.cat(out), EPSILON, 1.0)


def dpv_fuse(logp, dmaps, masks, d_candi, var=0.3):
    """(fused, log fused) as in BaseModel.forward_int, nmode default_upsample: models/models.py:663-672."""
    tofuse = gen_dpv_withmask(dmaps, masks, d_candi, var)
    fused = torch.exp(logp + torch.log(tofuse))
    fused = fused / torch.sum(fused, dim=1).unsqueeze(1)
    fused = torch.clamp(fused, EPSILON, 1.0)
    return fused, torch.log(fused)


def correlation(x1, x2, max_displacement=4):
    """81-channel mean-over-C correlation: models/correlation_native.py:13-23 (its own self-check pins
    it to the CUDA op at atol 1e-7, :64)."""
    B, C, H, W = x1.shape
    n = 2 * max_displacement + 1
    x2p = F.pad(x2, [max_displacement] * 4)
    cv = []
    for i in range(n):
        for j in range(n):
            cv.append(torch.mean(x1 * x2p[:, :, i:i + H, j:j + W], 1, keepdim=True))
    return torch.cat(cv, 1)


def correlation_general(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2):
    """The reference's correlation op for any configuration, restated from its kernel's index arithmetic in plain torch
    (autograd through this function is the oracle of the backward).

    models/correlation_package/correlation_cuda.cc:24-33 (output size: border = kernel radius + max_displacement,
    oH = ceil((H + 2 pad - 2 border) / stride1)); correlation_cuda_kernel.cu:57-59 (kernel_rad, displacement_rad =
    max_displacement / stride2, displacement_size), :62-63 (y1, x1 = output index * stride1 + max_displacement, in the
    coordinates of the zero-padded inputs of :30-38), :85-100 (sum over the kernel window and the channels of
    in1[y1+j, x1+i] * in2[y1 + tj*stride2 + j, x1 + ti*stride2 + i]), :107-109 (channel tc = (tj+dr)*ds + (ti+dr),
    divided by nelems = kernel_size^2 * C)."""
    B, C, H, W = x1.shape
    kr = (kernel_size - 1) // 2
    dr = max_displacement // stride2
    border = kr + max_displacement
    pH, pW = H + 2 * pad_size, W + 2 * pad_size
    oH = -(-(pH - 2 * border) // stride1)
    oW = -(-(pW - 2 * border) // stride1)
    # the sum reads padded rows y1 + tj*stride2 + j >= max_displacement - dr*stride2 - kr: non-negative for every
    # configuration the C ABI accepts; pad a little more so that plain slicing below never wraps
    extra = max(0, kr + dr * stride2 - max_displacement) + stride1
    a = F.pad(x1, [pad_size + extra] * 4)
    b = F.pad(x2, [pad_size + extra] * 4)
    ys = max_displacement + extra + stride1 * torch.arange(oH)
    xs = max_displacement + extra + stride1 * torch.arange(oW)
    out = []
    for tj in range(-dr, dr + 1):
        for ti in range(-dr, dr + 1):
            acc = 0
            for j in range(-kr, kr + 1):
                for i in range(-kr, kr + 1):
                    p1 = a[:, :, (ys + j)[:, None], (xs + i)[None, :]]
                    p2 = b[:, :, (ys + tj * stride2 + j)[:, None], (xs + ti * stride2 + i)[None, :]]
                    acc = acc + (p1 * p2).sum(1, keepdim=True)
            out.append(acc / float(kernel_size * kernel_size * C))
    return torch.cat(out, 1)


def inverse_warp(img, depth, pose_mat, intrinsics, mode="bilinear"):
    """Depth-driven inverse warp for a [B,4,4] or [B,3,4] pose -> (warped, valid).  Plain torch ops: autograd through
    this function is the oracle of the backward (losses/loss_blocks.py:116,151 call it under autograd).

    utils/inverse_warp.py:174-210: pixel2cam (:26-40), K @ pose (:200), cam2pixel (:43-69: Z clamped at 1e-3,
    normalisation with (w-1)/(h-1)), F.grid_sample with default align_corners, validity = |grid| <= 1 (:208).
    """
    b, _, h, w = img.shape
    i_range = torch.arange(0, h).view(1, h, 1).expand(1, h, w).type_as(depth)
    j_range = torch.arange(0, w).view(1, 1, w).expand(1, h, w).type_as(depth)
    pix = torch.stack((j_range, i_range, torch.ones(1, h, w).type_as(depth)), dim=1)
    cam = torch.matmul(intrinsics.inverse(), pix.expand(b, 3, h, w).reshape(b, 3, -1)).reshape(b, 3, h, w)
    cam = cam * depth.unsqueeze(1)
    proj = torch.matmul(intrinsics, pose_mat[:, 0:3, :])
    pc = torch.matmul(proj[:, :, :3], cam.reshape(b, 3, -1)) + proj[:, :, -1:]
    Z = pc[:, 2].clamp(min=1e-3)
    grid = torch.stack([2 * (pc[:, 0] / Z) / (w - 1) - 1, 2 * (pc[:, 1] / Z) / (h - 1) - 1], dim=2).reshape(b, h, w, 2)
    out = F.grid_sample(img, grid, padding_mode="zeros", mode=mode, align_corners=False)
    return out, grid.abs().max(dim=-1)[0] <= 1
