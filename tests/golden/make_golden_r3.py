"""Round-3 golden fixture g18: the tensors INSIDE one feedback step of the real reference model, so that the HIP ops of
that step can be checked on the reference's own inputs -- with the 3-D convolutions (MIOpen vs mkldnn: 2-3e-4 of depth
end to end) out of the picture.  Build container only (imports /root/reference, which never travels):

    python tests/golden/make_golden_r3.py

Two chained frames of BaseModel(nmode=default_feedback) at a 256x256 image (64x64 sweep: the smallest the encoder's pooling pyramid accepts), seeded inputs and weights.  Of
the SECOND frame (the one with a fed-back prev_output) the fixture stores
  * models/models.py:616-625  warp_feature: its input `feat_raw` and its output `warped_features`;
  * models/models.py:686-694  BV_cur, BV_resi (the residual of the 3-D network) and BV_cur_upd = log_softmax(BV_cur + BV_resi);
  * models/models.py:351      the decoder's tensor in front of its log_softmax (a crop) and the log-DPV it returns;
  * trainer/default_trainer.py:230-233  dpv_to_depthmap of BV_cur_upd and of the decoder's log-DPV (the crop).
Nothing of the reference is copied: its modules are imported and called; the two tensors that exist only inside functions
are caught by wrapping `warp_feature` / `F.log_softmax` for the duration of the call.  Data only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _import_reference  # noqa: E402,F401  (also sets sys.path for reference + repo)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

CROP = (slice(100, 132), slice(96, 160))   # rows, columns of the full-resolution crop that is stored


def main():
    homo, view, img_utils = _import_reference()
    import pdepth_amd
    from pdepth_amd import synth as S
    from util_host import cpu_vendor
    import models.get_model as gm
    import models.models as ref_models
    meta = dict(meta_torch=torch.__version__, meta_cpu_capability=torch.backends.cpu.get_cpu_capability(),
                meta_cpu_vendor=cpu_vendor(), meta_blas_mode=np.int32(pdepth_amd._native.host_blas_mode()))
    torch.nn.Module.cuda = lambda self, *a, **k: self     # models.py:399 calls .cuda() on the residual blocks
    cfg = S.default_cfg("default_feedback")
    torch.manual_seed(0)
    model = gm.get_model(cfg, 0)
    S.seed_weights(model, seed=18)
    model.eval()

    caught = {}
    orig_warp = ref_models.warp_homo.warp_feature
    orig_lsm = F.log_softmax

    def warp_spy(feat, *a, **k):
        out = orig_warp(feat, *a, **k)
        caught["feat_raw"], caught["warped"] = feat.detach().clone(), out.detach().clone()
        return out

    def lsm_spy(x, dim=None, **k):
        caught.setdefault("lsm_inputs", []).append(x.detach())
        return orig_lsm(x, dim=dim, **k)

    prev = None
    with torch.no_grad():
        for frame in range(2):
            inp = S.make_model_input(18000 + frame, B=1, V=1, H=256, W=256, D=64, pose="mono")
            inp["prev_output"] = prev
            if frame == 1:
                ref_models.warp_homo.warp_feature = warp_spy
                F.log_softmax = lsm_spy
            try:
                if frame == 0:
                    o = model([inp])[0]
                else:   # models.py:686-699, statement by statement, so that BV_resi can be kept
                    BV_cur, cost_volumes, last_features, first_features, warped_features = model.forward_exp(inp)
                    last_features.append(inp["rgb"][:, -1, :, :, :])
                    prev_output = inp["prev_output"].unsqueeze(1)
                    comb_volume = torch.cat([BV_cur.unsqueeze(1), prev_output, warped_features], dim=1)
                    BV_resi = model.based_3d(comb_volume, prob=False)
                    BV_upd = F.log_softmax(BV_cur + BV_resi, dim=1)
                    n_before = len(caught["lsm_inputs"])
                    BV_refined = model.base_decoder(torch.exp(BV_upd), img_features=last_features)
                    dec_pre = caught["lsm_inputs"][n_before]          # the decoder's only log_softmax (models.py:351)
                    o = {"output": [BV_cur, BV_upd], "output_refined": [BV_refined]}
            finally:
                ref_models.warp_homo.warp_feature = orig_warp
                F.log_softmax = orig_lsm
            prev = F.interpolate(o["output_refined"][-1].detach(), scale_factor=0.25, mode="nearest")  # default_trainer.py:221
        # the whole model on the same frame must give the same answer as the statements above
        inp2 = S.make_model_input(18001, B=1, V=1, H=256, W=256, D=64, pose="mono")
        inp2["prev_output"] = inp["prev_output"]
        o2 = model([inp2])[0]
        assert torch.equal(o2["output"][1], BV_upd) and torch.equal(o2["output_refined"][0], BV_refined)
        assert dec_pre.shape == BV_refined.shape and torch.equal(orig_lsm(dec_pre, dim=1), BV_refined)
        depth_low = img_utils.dpv_to_depthmap(BV_upd[0:1], inp["d_candi"], BV_log=True)
        depth_ref = img_utils.dpv_to_depthmap(BV_refined[0:1], inp["d_candi"], BV_log=True)
    ys, xs = CROP
    np.savez_compressed(
        os.path.join(HERE, "g18_feedback_step.npz"), input_seed=np.int32(18001), image_hw=np.int32([256, 256]),
        feat_raw=caught["feat_raw"].numpy(), warped_even=caught["warped"].numpy()[:, :, ::2],   # (outputs: every 2nd plane)
        BV_cur=BV_cur.numpy(), BV_resi=BV_resi.numpy(), BV_upd_even=BV_upd.numpy()[:, ::2], depth_low=depth_low.numpy(),
        crop=np.int32([ys.start, ys.stop, xs.start, xs.stop]),
        dec_pre_crop=dec_pre.numpy()[:, :, ys, xs], dec_logp_crop=BV_refined.numpy()[:, :, ys, xs],
        depth_ref_crop=depth_ref.numpy()[:, ys, xs], **meta)
    print("g18_feedback_step.npz", os.path.getsize(os.path.join(HERE, "g18_feedback_step.npz")), "bytes;",
          "BV_resi range", float(BV_resi.min()), float(BV_resi.max()), "depth_low range", float(depth_low.min()), float(depth_low.max()))


if __name__ == "__main__":
    main()
