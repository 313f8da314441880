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
