"""Drop-in for the hot-path functions of the reference's utils/img_utils.py."""
import numpy as np

from .. import ops


def dpv_to_depthmap(dpv, d_candi, BV_log=False):
    """E[d] of a [1,D,H,W] (log-)DPV -> [1,H,W]  (utils/img_utils.py:52-61)."""
    if dpv.shape[0] != 1:
        raise Exception("Unable to handle this case")
    return ops.dpv_expect(dpv, d_candi, BV_log=BV_log)


def powerf(d_min, d_max, nDepth, power):
    """Depth candidates, float64 (utils/img_utils.py:80-85)."""
    x = np.power(np.linspace(start=0, stop=1, num=nDepth), power)
    return np.array([d_min + (d_max - d_min) * v for v in x])


def gen_ufield(dpv_predicted, d_candi, intr_up, visualizer=None, img=None, BV_log=True, normalize=False, mask=None,
               cfg=None, cfgx=None):
    """Uncertainty field of a [1,D,H,W] (log-)DPV -> (plane [1,D,W], masked depth map [1,H,W]).

    Same signature and parameter branches as utils/img_utils.py:268-358: cfgx = {"unc_ang", "unc_shift", "unc_span"}
    (min depth 3, quash on), or cfg.data.dataset_path containing "kitti" (5 rows, band [0.6, 0.9], no quash) or "ilim"
    (no shift, band [1.0, 1.3], min depth 3, quash).  visualizer / img are accepted and unused, like in the reference."""
    if dpv_predicted.shape[0] != 1:
        raise Exception("Unable to handle this case")
    if cfgx is not None:
        pshift, zstart, zend, mind, quash = cfgx["unc_ang"], cfgx["unc_shift"], cfgx["unc_shift"] + cfgx["unc_span"], 3., True
    elif "kitti" in cfg.data.dataset_path:
        pshift, zstart, zend, mind, quash = 5, 0.6, 0.6 + 0.3, 0., False
    elif "ilim" in cfg.data.dataset_path:
        pshift, zstart, zend, mind, quash = 0, 1.0, 1.0 + 0.3, 3., True
    else:
        raise UnboundLocalError("gen_ufield: dataset_path names neither kitti nor ilim")  # the reference fails the same way
    plane, depth_zero = ops.ufield(dpv_predicted, d_candi, intr_up.reshape(1, 3, 3), mask, BV_log=BV_log, unc_ang=pshift,
                                   z_start=zstart, z_end=zend, min_depth=mind, quash=quash)
    if normalize:
        minval, _ = plane.min(1)
        maxval, _ = plane.max(1)
        plane = (plane - minval) / (maxval - minval)
    return plane, depth_zero


def compute_unc_field(dpv_refined_predicted, dpv_refined_truth, d_candi, intr_refined, mask_refined, cfg):
    """(field of the ground-truth DPV, field of the predicted log-DPV, masked depth map of the prediction)
    (utils/img_utils.py:178-181; called by the evaluation loop, trainer/default_trainer.py:243-244): two gen_ufield
    collapses through the dataset branch of `cfg` -- the truth as probabilities under its validity mask, the prediction as
    a log-DPV without one.  intr_refined [1,3,3]."""
    unc_field_truth, _ = gen_ufield(dpv_refined_truth, d_candi, intr_refined.squeeze(0), BV_log=False, mask=mask_refined, cfg=cfg)
    unc_field_predicted, debugmap = gen_ufield(dpv_refined_predicted, d_candi, intr_refined.squeeze(0), BV_log=True, cfg=cfg)
    return unc_field_truth, unc_field_predicted, debugmap


def compute_unc_rmse(unc_field_truth, unc_field_predicted, d_candi, plot=False):
    """Error between two [1,D,W] uncertainty fields (utils/img_utils.py:183-202): E[d] per column of each, the
    predicted one zeroed in the first and last column, columns where either is NaN (no qualifying pixel) dropped.
    Despite the name the value returned is the mean absolute difference -- the reference overwrites its RMSE with it
    (:192-193).  `plot` is accepted and ignored (the reference draws the two curves with matplotlib)."""
    import torch
    truth_depth = dpv_to_depthmap(unc_field_truth.unsqueeze(2), d_candi, BV_log=False).squeeze(0).squeeze(0)
    pred_depth = dpv_to_depthmap(unc_field_predicted.unsqueeze(2), d_candi, BV_log=False).squeeze(0).squeeze(0)
    pred_depth[0] = 0
    pred_depth[-1] = 0
    usable = ~torch.isnan(truth_depth) & ~torch.isnan(pred_depth)
    truth_depth = torch.where(usable, truth_depth, torch.zeros_like(truth_depth))
    pred_depth = torch.where(usable, pred_depth, torch.zeros_like(pred_depth))
    return torch.sum(torch.abs(truth_depth - pred_depth)) / torch.sum(usable)
