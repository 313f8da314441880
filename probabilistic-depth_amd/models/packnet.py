"""Host object with the contract of the reference's ``PacknetModel`` (models/packnet.py:304-406, made by
``get_model`` for ``model_name == 'packnet'``: models/get_model.py:9-10).

What the reference's class does that belongs to this package is its head: for every batch item ``est_swp_volume_v4`` on
the feature maps, ``torch.cat``, ``F.log_softmax(dim=1)`` (:362-394).  That is ``PacknetHead`` -- one fused launch for
the whole batch, sources straight from the encoder output into the sweep's staging layout.  The PackNet encoder /
decoder themselves (packnet.py:15-302: a dense 2-D CNN with packing / unpacking blocks) are out of scope
 and are NOT provided: this class takes them as modules with the reference's call
signatures and raises when they are absent, so that a user of the reference who has the network can plug it in and a user
who has not gets a sentence instead of wrong numbers.

    base_encoder(rgb [N,3,H,W]) -> (feats_raw: list of [N,c_i,h_i,w_i], feat_imgs [N,C,h,w])      packnet.py:347
    base_decoder(dpv [B,D,h,w], features of the reference view: list) -> log-DPV [B,D,H',W']       packnet.py:402
"""
import torch
import torch.nn as nn

from .packnet_head import PacknetHead


class _FeatImgs(nn.Module):
    """the encoder as PacknetHead calls it (rgb -> feature maps), remembering the raw feature pyramid of the call"""

    def __init__(self, owner):
        super().__init__()
        self._owner = [owner]   # (a list: not registered as a sub-module, no reference cycle in the module tree)
        self.feats_raw = None

    def forward(self, rgb):
        self.feats_raw, feat_imgs = self._owner[0].base_encoder(rgb)
        return feat_imgs


class PacknetModel(nn.Module):
    def __init__(self, cfg, id, base_encoder=None, base_decoder=None):
        super().__init__()
        self.cfg = cfg
        self.id = id
        self.sigma_soft_max = cfg.var.sigma_soft_max
        self.feature_dim = cfg.var.feature_dim
        self.nmode = cfg.var.nmode
        self.base_encoder = base_encoder
        self.base_decoder = base_decoder
        self.head = PacknetHead(cfg, encoder=_FeatImgs(self))

    def attach_networks(self, base_encoder, base_decoder):
        self.base_encoder, self.base_decoder = base_encoder, base_decoder
        return self

    def _need(self, what):
        if getattr(self, what) is None:
            raise NotImplementedError(
                f"packnet: the PackNet network is not part of this package (SURVEY.md section 2); pass {what} to PacknetModel / "
                "attach_networks -- the sweep + DPV head runs without it: models.packnet_head.PacknetHead")

    def _inference_only(self):
        """The sweep kernels have no backward: this host object is the reference's model for `--eval`.  The reference class is
        trainable; a caller who runs this one with autograd on and trainable networks attached is told so, once."""
        if torch.is_grad_enabled() and not getattr(self, "_warned_no_grad", False) and any(p.requires_grad for p in self.parameters()):
            import warnings
            warnings.warn("pdepth_amd PacknetModel runs under torch.no_grad(): the fused sweep + DPV head has no backward; the attached "
                          "networks receive no gradients (inference / --eval only)", RuntimeWarning, stacklevel=3)
            self._warned_no_grad = True

    def forward_encoder(self, model_input):
        """-> (BV = log_softmax(cost volumes) [B,D,h,w], feature_set[view] = list of raw feature maps [B,c_i,h_i,w_i])"""
        self._inference_only()
        with torch.no_grad():
            return self._forward_encoder(model_input)

    def _forward_encoder(self, model_input):
        self._need("base_encoder")
        rgb = model_input["rgb"]
        BV, _depth = self.head(model_input)
        feature_set = [[] for _ in range(rgb.shape[1])]
        for feat_raw in self.head.encoder.feats_raw:   # packnet.py:359-367
            r = feat_raw.view(rgb.shape[0], rgb.shape[1], feat_raw.shape[1], feat_raw.shape[2], feat_raw.shape[3])
            for tw in range(rgb.shape[1]):
                feature_set[tw].append(r[:, tw])
        self.head.encoder.feats_raw = None
        return BV, feature_set

    def forward(self, input):
        self._inference_only()
        with torch.no_grad():
            BV_cur, feature_set = self._forward_encoder(input)
            self._need("base_decoder")
            BV_cur_refined = self.base_decoder(torch.exp(BV_cur), feature_set[-1])
        return {"output": [BV_cur], "output_refined": [BV_cur_refined], "flow": None, "flow_refined": None}
