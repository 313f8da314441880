"""Round-4 golden fixture g22: the tensors inside EVERY feedback step of a five-frame window of the real reference model
(BASELINE config 4), so that the HIP ops of the feedback path are checked on the reference's own inputs frame by frame --
teacher forcing: the 3-D convolutions (MIOpen vs mkldnn, 2-3e-4 m end to end) never enter.  Build container only (imports
/root/reference, which never travels):

    python tests/golden/make_golden_r4.py

Five chained frames of BaseModel(nmode=default_feedback) at a 256x256 image (64x64 sweep), seeded inputs and weights, frames
1..4 with a fed-back prev_output (trainer/default_trainer.py:221).  Per such frame the fixture stores (crops / subsets keep
it small; the ops are per pixel, or -- warp_feature -- per channel):
  * models/models.py:616-625   warp_feature: channels 0, 4, 8, ... of `feat_raw` (channel i is warped with plane i only, so the
                               op on that subset with d_candi[::4] is the reference's output restricted to it) and the same
                               channels of `warped_features` (rows ROWS);
  * models/models.py:686-694   BV_cur, BV_resi (rows ROWS), BV_cur_upd = log_softmax(BV_cur + BV_resi) (even planes), its depth;
  * models/models.py:351       the decoder's tensor in front of its log_softmax (a crop), the log-DPV it returns (even planes),
                               its depth (trainer/default_trainer.py:230-233).
Nothing of the reference is copied: its modules are imported and called; tensors that exist only inside functions are caught
by wrapping `warp_feature` / `F.log_softmax` for the duration of the call.  Data only.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _import_reference  # noqa: E402,F401  (also sets sys.path for reference + repo)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

ROWS = slice(24, 40)                       # rows of the 64x64 low-resolution tensors that are stored
CROP = (slice(104, 120), slice(96, 160))   # rows, columns of the full-resolution crop that is stored
NFRAMES = 5


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
    S.seed_weights(model, seed=22)
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

    out = {}
    ys, xs = CROP
    prev = None
    with torch.no_grad():
        for frame in range(NFRAMES):
            inp = S.make_model_input(22000 + frame, B=1, V=1, H=256, W=256, D=64, pose="mono")
            inp["prev_output"] = prev
            if frame == 0:
                o = model([inp])[0]
            else:
                ref_models.warp_homo.warp_feature = warp_spy
                F.log_softmax = lsm_spy
                caught.clear()
                try:   # models.py:686-699, statement by statement, so that BV_resi can be kept
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
                # the whole model on the same frame gives the same answer as the statements above
                o2 = model([inp])[0]
                assert torch.equal(o2["output"][1], BV_upd) and torch.equal(o2["output_refined"][0], BV_refined)
                depth_low = img_utils.dpv_to_depthmap(BV_upd[0:1], inp["d_candi"], BV_log=True)
                depth_ref = img_utils.dpv_to_depthmap(BV_refined[0:1], inp["d_candi"], BV_log=True)
                f = "f%d_" % frame
                out[f + "feat_raw_c4"] = caught["feat_raw"].numpy()[:, :, ::4]
                out[f + "warped_c4"] = caught["warped"].numpy()[:, :, ::4, ROWS]
                out[f + "BV_cur"] = BV_cur.numpy()[:, :, ROWS]
                out[f + "BV_resi"] = BV_resi.numpy()[:, :, ROWS]
                out[f + "BV_upd_even"] = BV_upd.numpy()[:, ::2, ROWS]
                out[f + "depth_low"] = depth_low.numpy()[:, ROWS]
                out[f + "dec_pre_crop"] = dec_pre.numpy()[:, :, ys, xs]
                out[f + "dec_logp_even"] = BV_refined.numpy()[:, ::2, ys, xs]
                out[f + "depth_ref_crop"] = depth_ref.numpy()[:, ys, xs]
                print("frame", frame, "BV_resi range", float(BV_resi.min()), float(BV_resi.max()),
                      "depth_low", float(depth_low.min()), float(depth_low.max()), flush=True)
            prev = F.interpolate(o["output_refined"][-1].detach(), scale_factor=0.25, mode="nearest")  # default_trainer.py:221
    path = os.path.join(HERE, "g22_feedback_window.npz")
    np.savez_compressed(path, input_seed0=np.int32(22000), nframes=np.int32(NFRAMES), image_hw=np.int32([256, 256]),
                        rows=np.int32([ROWS.start, ROWS.stop]), crop=np.int32([ys.start, ys.stop, xs.start, xs.stop]), **out, **meta)
    print("g22_feedback_window.npz", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
