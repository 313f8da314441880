"""The sweep + DPV head of the reference's PackNet model, fused (models/packnet.py:343-396).

``PacknetModel.forward_encoder`` ends in the only place where the reference chains the plane sweep, the softmax over
depth and the expectation without anything in between: for every batch item ``est_swp_volume_v4`` on the feature maps
(:366-388), ``torch.cat``, then ``F.log_softmax(cost_volumes, dim=1)`` (:394) -- and its consumers regress the depth
with ``dpv_to_depthmap``.  The PackNet encoder / decoder themselves are a separate dense CNN and out of this package's
scope; this module is that head with the same inputs and the same result, as ONE launch of
the fused kernel (``ops.sweep_dpv``: the cost volume never reaches HBM) for the whole batch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class PacknetHead(nn.Module):
    """feature maps + model_input -> log-DPV [B,D,h,w] (and the expected depth [B,h,w] as a by-product).

    ``encoder`` is any module mapping rgb [N,3,H,W] -> feature maps [N,C,h,w] (PackNet's in the reference; pass None to
    feed precomputed features).  Reference view = LAST view, sources = the others (packnet.py:372-373)."""

    def __init__(self, cfg, encoder=None):
        super().__init__()
        self.sigma_soft_max = cfg.var.sigma_soft_max
        self.encoder = encoder
        self.sweep_algo = "auto"
        self.sweep_blas = None

    def features(self, rgb):
        """rgb [B,V+1,3,H,W] -> [B,V+1,C+3,h,w]: encoder features + the average-pooled image (packnet.py:348-357)."""
        B, V1 = rgb.shape[0], rgb.shape[1]
        flat = rgb.reshape(B * V1, rgb.shape[2], rgb.shape[3], rgb.shape[4])
        feat = self.encoder(flat)
        rate = int(flat.shape[3] / feat.shape[3])
        both = torch.cat((feat, F.avg_pool2d(flat, rate)), dim=1)
        return both.view(B, V1, both.shape[1], both.shape[2], both.shape[3])

    @torch.no_grad()
    def forward(self, model_input, feat_imgs_all=None, want_cost=False):
        """Returns (BV = log_softmax(cost volumes, dim=1) [B,D,h,w], depth [B,h,w][, cost volumes])."""
        ref = src = None
        if feat_imgs_all is None and self.sweep_algo == "auto":
            # the encoder epilogue kernel: pooled image appended, source views straight into the sweep's staging layout
            rgb = model_input["rgb"]
            B, V1 = rgb.shape[0], rgb.shape[1]
            flat = rgb.reshape(B * V1, rgb.shape[2], rgb.shape[3], rgb.shape[4])
            enc = self.encoder(flat).float()   # (outside the try: an encoder failure is not a reason to run it twice)
            try:
                src, ref = ops.pack_views(enc, flat.float(), V1, len(model_input["d_candi"]))
            except ops.UnsupportedShape:   # a shape the packed sweep does not take; other native failures propagate
                src = ref = None
                # the encoder's output is reused (the fallback used to run the encoder a second time)
                rate = int(flat.shape[3] / enc.shape[3])
                both = torch.cat((enc, F.avg_pool2d(flat.float(), rate)), dim=1)
                feat_imgs_all = both.view(B, V1, both.shape[1], both.shape[2], both.shape[3])
        if src is None:
            if feat_imgs_all is None:
                feat_imgs_all = self.features(model_input["rgb"])
            ref, src = feat_imgs_all[:, -1], feat_imgs_all[:, :-1]
            if self.sweep_algo == "auto":   # precomputed features: re-laid once, then the packed entry
                try:
                    src = ops.pack_source(src, len(model_input["d_candi"]))
                except ops.UnsupportedShape:
                    pass
        poses = model_input["src_cam_poses"].float()
        K = model_input["intrinsics"].float()
        cost, BV, depth = ops.sweep_dpv(
            ref, src, K, poses[:, :-1, :3, :3], poses[:, :-1, :3, 3],
            model_input["unit_ray"].float(), K[:, :2, 2].contiguous(), model_input["d_candi"], self.sigma_soft_max,
            feat_dist="L2", algo=self.sweep_algo, want_cost=want_cost, blas=self.sweep_blas)   # (a PackedSource implies "auto")
        return (BV, depth, cost) if want_cost else (BV, depth)
