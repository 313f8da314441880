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
