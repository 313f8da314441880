"""Batched entry points of the hot path (thin, typed wrappers over ``_native``).

These are what the host model and the harness call; the reference-compatible per-item
functions in ``warping.homography`` / ``utils.img_utils`` are wrappers around them.
All tensors are fp32 device tensors; nothing here runs on the CPU.
"""
from __future__ import annotations

import torch

from . import _native

METRICS = {"L2": _native.METRIC_L2, "L1": _native.METRIC_L1}
# "tiled1" / "tiled2" / "cells" force one of the implementations behind "auto" (parity tests, A/B timing)
ALGOS = {"auto": _native.ALGO_AUTO, "direct": _native.ALGO_DIRECT, "tiled1": _native.ALGO_TILED_1,
         "tiled2": _native.ALGO_TILED_2, "cells": _native.ALGO_CELLS, "mfma": _native.ALGO_MFMA, "corr": _native.ALGO_CORR, "dist": _native.ALGO_DIST}
BLAS_MODES = {"fma": _native.BLAS_FMA, "separate": _native.BLAS_SEPARATE, None: None}


def _metric(feat_dist):
    if feat_dist not in METRICS:
        # same error as warping/homography.py:133
        raise Exception("undefined metric for feature distance ...")
    return METRICS[feat_dist]


_D_CANDI_CACHE = {}


def d_candi_tensor(d_candi, device):
    """float64 numpy / list / tensor -> fp32 device tensor (homography.py:115 cast).

    Host arrays are uploaded once per (values, device) and cached: the candidates are a constant of a run, and a
    pageable host-to-device copy per op would synchronise the stream (and cannot be captured in a HIP graph)."""
    if isinstance(d_candi, torch.Tensor):
        return d_candi.to(device=device, dtype=torch.float32)
    import numpy as np
    arr = np.ascontiguousarray(np.asarray(d_candi), dtype=np.float32)
    dev = torch.device(device)
    key = (arr.tobytes(), dev.type, dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else -1))
    t = _D_CANDI_CACHE.get(key)
    if t is None:
        if len(_D_CANDI_CACHE) > 64:
            _D_CANDI_CACHE.clear()
        t = torch.from_numpy(arr.copy()).to(dev)
        _D_CANDI_CACHE[key] = t
    return t


def sweep_cost(ref, src, K, R, t, rays, cxcy, d_candi, sigma, feat_dist="L2", algo="auto", blas=None):
    """cost [B,D,H,W].  Batched est_swp_volume_v4 (warping/homography.py:98-135).

    blas: None = reproduce the rounding of THIS host's CPU BLAS (see _native.host_blas_mode),
    "fma" / "separate" to force one (golden fixtures record the mode of the host that made them).
    """
    cost, _, _ = _native.sweep(ref, src, K, R, t, rays, cxcy, d_candi_tensor(d_candi, ref.device), sigma,
                               _metric(feat_dist), ALGOS[algo], want_cost=True, blas_mode=BLAS_MODES[blas])
    return cost


def sweep_dpv(ref, src, K, R, t, rays, cxcy, d_candi, sigma, feat_dist="L2", algo="auto",
              want_cost=False, want_logp=True, want_depth=True, blas=None):
    """Fused sweep -> log_softmax(dim=1) -> E[d].  Returns (cost|None, logp|None, depth|None).

    models/packnet.py:380-394 + utils/img_utils.py:52-61 in one kernel.
    """
    return _native.sweep(ref, src, K, R, t, rays, cxcy, d_candi_tensor(d_candi, ref.device), sigma,
                         _metric(feat_dist), ALGOS[algo], want_cost=want_cost, want_logp=want_logp,
                         want_depth=want_depth, blas_mode=BLAS_MODES[blas])


UnsupportedShape = _native.UnsupportedShape


def pack_source(src, n_planes=64, algo="auto", feat_dist="L2"):
    """Source views [B,V,C,H,W] -> the sweep kernels' staging layout, once (pdepth_pack_source_f32); pass the result as
    `src` to sweep_cost / sweep_dpv with the same algo and feat_dist (the kernel they select decides whether the layout is
    mean-centred).  The re-layout is 10 % of a fused sweep call.  Raises UnsupportedShape for shapes the packed sweep does
    not take."""
    return _native.pack_source(src, n_planes, ALGOS[algo], _metric(feat_dist))


def pack_views(feat, rgb, n_views, n_planes=64):
    """Encoder epilogue: cat(feat, avg_pool2d(rgb)) -> (packed source views, NCHW reference view) in one pass
    (pdepth_pack_views_f32; models/models.py:518-534).  Pass the PackedSource as `src` and the tensor as `ref` to
    sweep_cost / sweep_dpv.  Raises UnsupportedShape for shapes the packed sweep does not take (callers use cat + avg_pool2d)."""
    return _native.pack_views(feat, rgb, n_views, n_planes)


def dpv_reduce(logits, d_candi, want_logp=True, want_depth=True, inplace=False):
    """(logp, depth) from logits [B,D,H,W]: log_softmax(dim=1) + dpv_to_depthmap(BV_log=True).

    d_candi may be None when only the log-softmax is wanted (want_depth=False).
    """
    if d_candi is None:
        if want_depth:
            raise RuntimeError("dpv_reduce: d_candi is required for the depth output")
        dc = torch.zeros(logits.shape[1], dtype=torch.float32, device=logits.device)
    else:
        dc = d_candi_tensor(d_candi, logits.device)
    return _native.dpv_reduce(logits, dc, want_logp, want_depth, inplace)


def dpv_reduce_ex(logits, d_candi=None, addend=None, want_logp=True, want_prob=False, want_depth=False, want_var=False,
                  want_quarter=False, inplace=False):
    """One pass over (logits [+ addend]): log_softmax over D and any of exp(logp), E[d], Var[d], the nearest
    quarter-resolution log-DPV (the next frame's prev_output).  Returns a dict with the requested outputs.

    models/models.py:694 + :697 (feedback update and decoder input), trainer/default_trainer.py:333-336 (variance),
    :221 (prev_output)."""
    if d_candi is None:
        if want_depth or want_var:
            raise RuntimeError("dpv_reduce_ex: d_candi is required for the depth / variance outputs")
        dc = torch.zeros(logits.shape[1], dtype=torch.float32, device=logits.device)
    else:
        dc = d_candi_tensor(d_candi, logits.device)
    return _native.dpv_reduce_ex(logits, dc, addend, want_logp, want_prob, want_depth, want_var, want_quarter, inplace)


def ufield(dpv, d_candi, intr, mask=None, BV_log=True, unc_ang=5, z_start=0.6, z_end=0.9, min_depth=0.0, quash=False):
    """Uncertainty-field collapse, batched: (plane [B,D,W], masked depth [B,H,W]) of a (log-)DPV [B,D,H,W]
    (utils/img_utils.py:268-358).  mask [B,H,W] | [B,1,H,W] | None."""
    if mask is not None and mask.dim() == 4:
        mask = mask[:, 0]
    # depth of rows shifted in from outside the image = dpv_to_depthmap of the zero padding = the fp32 sum of the candidates:
    # formed on the host when the candidates come from the host (no device synchronisation inside the call)
    d_sum = None
    if not isinstance(d_candi, torch.Tensor):
        import numpy as np
        d_sum = float(torch.from_numpy(np.ascontiguousarray(np.asarray(d_candi), dtype=np.float32)).sum())
    return _native.ufield(dpv, d_candi_tensor(d_candi, dpv.device), intr, mask, BV_log, unc_ang, z_start, z_end, min_depth, quash,
                          d_sum=d_sum)


def dpv_expect(dpv, d_candi, BV_log=False):
    """depth [B,H,W] from a (log-)DPV [B,D,H,W] (utils/img_utils.py:52-61, batched)."""
    return _native.dpv_expect(dpv, d_candi_tensor(d_candi, dpv.device), BV_log)


def dpv_moments(dpv, d_candi, BV_log=True):
    """(mean, variance) [B,H,W] of the depth distribution of a (log-)DPV (trainer/default_trainer.py:333-336)."""
    return _native.dpv_moments(dpv, d_candi_tensor(d_candi, dpv.device), BV_log)


def warp_feature(src, K, R, t, rays, cxcy, d_candi, blas=None):
    """[B,V,D,H,W] diagonal warp (warping/homography.py:137-168, batched)."""
    return _native.warp_feature(src, K, R, t, rays, cxcy, d_candi_tensor(d_candi, src.device),
                                blas_mode=BLAS_MODES[blas])


def sample_coords(K, R, t, rays, cxcy, d_candi, H, W, blas=None, algo=_native.ALGO_AUTO):
    return _native.sample_coords(K, R, t, rays, cxcy, d_candi_tensor(d_candi, K.device), H, W, algo=algo,
                                 blas_mode=BLAS_MODES[blas])


def dpv_fuse(logp, dmaps, masks, d_candi, var=0.3, eps=None, want_fused=True, want_log=True):
    """Bayesian fusion of a log-DPV with the Gaussian soft label of a sparse depth map.

    utils/img_utils.py:360-375 (gen_dpv_withmask) + models/models.py:666-672 in one kernel.
    masks may be [B,1,H,W] (reference layout, channel 0 is used) or [B,H,W].
    Returns (fused probabilities | None, log fused | None).
    """
    if eps is None:
        eps = torch.finfo(float).eps  # reference: utils/img_utils.py:12
    if masks.dim() == 4:
        masks = masks[:, 0]
    return _native.dpv_fuse(logp, dmaps.float(), masks.float(), d_candi_tensor(d_candi, logp.device), var, eps,
                            want_fused, want_log)


class _CorrelationFn(torch.autograd.Function):
    """Autograd binding like models/correlation_package/correlation.py:6-44 (CorrelationFunction)."""

    @staticmethod
    def forward(ctx, x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply):
        ctx.save_for_backward(x1, x2)
        ctx.cfg = (pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
        return _native.correlation_forward(x1, x2, *ctx.cfg)

    @staticmethod
    def backward(ctx, grad_out):
        x1, x2 = ctx.saved_tensors
        g1, g2 = _native.correlation_backward(x1, x2, grad_out, *ctx.cfg, want1=ctx.needs_input_grad[0],
                                              want2=ctx.needs_input_grad[1])
        return g1, g2, None, None, None, None, None, None


def correlation(x1, x2, pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1, corr_multiply=1):
    """The reference's native correlation op, forward and backward (models/correlation_package/correlation.py:6-61)."""
    if x1.requires_grad or x2.requires_grad:
        return _CorrelationFn.apply(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
    return _native.correlation_forward(x1, x2, pad_size, kernel_size, max_displacement, stride1, stride2, corr_multiply)
