"""MI355X-native plane-sweep / DPV hot path of soulslicer/probabilistic-depth.

Package layout mirrors the reference modules it stands in for:

    warping.homography   est_swp_volume_v4, warp_feature      (reference warping/homography.py)
    warping.view         unit-ray table                        (reference warping/view.py)
    utils.img_utils      dpv_to_depthmap, powerf               (reference utils/img_utils.py)
    models.get_model     get_model(cfg, id)                    (reference models/get_model.py)
    ops                  batched entry points over the C ABI   (include/pdepth.h)

The directory name contains a hyphen; import it as ``pdepth_amd`` (alias module at the repo
root) or with ``importlib.import_module("probabilistic-depth_amd")``.
"""
from . import _native  # noqa: F401
from . import ops  # noqa: F401

__all__ = ["_native", "ops"]
