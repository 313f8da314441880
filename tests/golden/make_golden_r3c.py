"""Round-3 golden fixtures g20 and g21.  g20: the reference's BaseModel in nmode=default_upsample (models/models.py:658-678: the DPV of the
sweep fused with the Gaussian soft label of a sparse depth map, img_utils.gen_dpv_withmask) -- the one nmode fixture g8 does
not hold.  Build container only (imports /root/reference):

    python tests/golden/make_golden_r3c.py

Seeded inputs (synth.make_model_input + a seeded sparse depth map and mask) and weights; stored: sub-sampled fused / plain
low-resolution log-DPVs and the three depth maps.

g21: BaseModel nmode=default on a batch of TWO items with TWO source views each (the reference loops over the items and, per
item, over the views: models/models.py:522-545, warping/homography.py:124-131; this package makes one batched call) -- the
same captures per item.  Data only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _import_reference  # noqa: E402,F401

import numpy as np  # noqa: E402
import torch  # noqa: E402

SEED_INPUT, SEED_SPARSE, SEED_WEIGHTS = 20000, 20, 20


def sparse_depth(seed=SEED_SPARSE, B=1, h=64, w=64):
    """A LIDAR-like sparse depth map at the sweep resolution: 30 % of the pixels carry a depth in [6, 36] m."""
    g = torch.Generator().manual_seed(seed)
    masks = (torch.rand(B, 1, h, w, generator=g) > 0.7).float()
    dmaps = (torch.rand(B, h, w, generator=g) * 30 + 6) * masks[:, 0]
    return dmaps, masks


def main():
    homo, view, img_utils = _import_reference()
    import pdepth_amd
    from pdepth_amd import _native, synth as S
    from util_host import cpu_vendor
    import models.get_model as gm
    meta = dict(meta_torch=torch.__version__, meta_cpu_capability=torch.backends.cpu.get_cpu_capability(),
                meta_cpu_vendor=cpu_vendor(), meta_blas_mode=np.int32(_native.host_blas_mode()))
    torch.nn.Module.cuda = lambda self, *a, **k: self
    cfg = S.default_cfg("default_upsample")
    torch.manual_seed(0)
    model = gm.get_model(cfg, 0)
    S.seed_weights(model, seed=SEED_WEIGHTS)
    model.eval()
    inp = S.make_model_input(SEED_INPUT, B=1, V=1, H=256, W=256, D=64, pose="mono")
    inp["dmaps"], inp["masks"] = sparse_depth()
    with torch.no_grad():
        out = model([inp])[0]
    fused, plain = out["output"]
    refined = out["output_refined"][0]
    d = inp["d_candi"]
    np.savez_compressed(
        os.path.join(HERE, "g20_upsample_model.npz"),
        seeds=np.int32([SEED_INPUT, SEED_SPARSE, SEED_WEIGHTS]),
        fused_logdpv_sub=fused.numpy()[:, ::2, ::2, ::2], plain_logdpv_sub=plain.numpy()[:, ::4, ::2, ::2],
        depth_fused=img_utils.dpv_to_depthmap(fused, d, BV_log=True).numpy(),
        depth_plain=img_utils.dpv_to_depthmap(plain, d, BV_log=True).numpy(),
        depth_refined=img_utils.dpv_to_depthmap(refined, d, BV_log=True).numpy(), **meta)
    print("g20_upsample_model.npz", os.path.getsize(os.path.join(HERE, "g20_upsample_model.npz")), "bytes;",
          "fused depth range", float(img_utils.dpv_to_depthmap(fused, d, BV_log=True).min()), float(img_utils.dpv_to_depthmap(fused, d, BV_log=True).max()))


def main_b2v2():
    homo, view, img_utils = _import_reference()
    import pdepth_amd
    from pdepth_amd import _native, synth as S
    from util_host import cpu_vendor
    import models.get_model as gm
    meta = dict(meta_torch=torch.__version__, meta_cpu_capability=torch.backends.cpu.get_cpu_capability(),
                meta_cpu_vendor=cpu_vendor(), meta_blas_mode=np.int32(_native.host_blas_mode()))
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.manual_seed(0)
    model = gm.get_model(S.default_cfg("default"), 0)
    S.seed_weights(model, seed=21)
    model.eval()
    inp = S.make_model_input(21000, B=2, V=2, H=256, W=256, D=64, pose="mono")
    with torch.no_grad():
        out = model([inp])[0]
    low, refined = out["output"][-1], out["output_refined"][-1]
    d = inp["d_candi"]
    depth_low = np.stack([img_utils.dpv_to_depthmap(low[b:b + 1], d, BV_log=True).numpy()[0] for b in range(2)])
    depth_ref = np.stack([img_utils.dpv_to_depthmap(refined[b:b + 1], d, BV_log=True).numpy()[0] for b in range(2)])
    np.savez_compressed(os.path.join(HERE, "g21_model_b2v2.npz"), seeds=np.int32([21000, 21]), logdpv_sub=low.numpy()[:, ::4, ::2, ::2],
                        depth_low=depth_low, depth_refined=depth_ref, **meta)
    print("g21_model_b2v2.npz", os.path.getsize(os.path.join(HERE, "g21_model_b2v2.npz")), "bytes; depth range", float(depth_low.min()), float(depth_low.max()))


if __name__ == "__main__":
    main()
    main_b2v2()
