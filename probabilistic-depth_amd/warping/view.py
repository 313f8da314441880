"""Unit-ray table (reference warping/view.py:16-62 + kittiloader/kitti.py:284-293,311-312).

Host-side, once per camera; evaluated in float64 exactly like the reference's per-pixel
Python loop, but vectorised.
"""
import math

import numpy as np


def pixel_to_ray(pixel, vfov=45, hfov=60, pixel_width=320, pixel_height=240):
    """(x, y, 1) ray through the centre of `pixel` (warping/view.py:16-30)."""
    x, y = pixel
    x_vect = math.tan(math.radians(hfov / 2.0)) * ((2.0 * ((x + 0.5) / pixel_width)) - 1.0)
    y_vect = math.tan(math.radians(vfov / 2.0)) * ((2.0 * ((y + 0.5) / pixel_height)) - 1.0)
    return (x_vect, y_vect, 1.0)


def normalised_pixel_to_ray_array(width=320, height=240, hfov=60, vfov=45, normalize_z=True):
    """[height, width, 3] float64 ray table (warping/view.py:32-62)."""
    xs = math.tan(math.radians(hfov / 2.0)) * ((2.0 * ((np.arange(width, dtype=np.float64) + 0.5) / width)) - 1.0)
    ys = math.tan(math.radians(vfov / 2.0)) * ((2.0 * ((np.arange(height, dtype=np.float64) + 0.5) / height)) - 1.0)
    rays = np.empty((height, width, 3), dtype=np.float64)
    rays[:, :, 0] = xs[None, :]
    rays[:, :, 1] = ys[:, None]
    rays[:, :, 2] = 1.0
    if not normalize_z:
        rays /= np.linalg.norm(rays, axis=2, keepdims=True)
    return rays


def camera_from_fov(width, height, hfov, vfov):
    """KITTI-branch intrinsics + ray table at the sweep resolution (kittiloader/kitti.py:284-320).

    Returns the reference's cam_intrinsic dict keys that the hot path reads.
    """
    import torch
    K = np.zeros((3, 3))
    K[2, 2] = 1.0
    K[0, 0] = (width / 2.0) / math.tan(math.radians(hfov / 2.0))
    K[0, 2] = width / 2.0
    K[1, 1] = (height / 2.0) / math.tan(math.radians(vfov / 2.0))
    K[1, 2] = height / 2.0
    rays = normalised_pixel_to_ray_array(width=width, height=height, hfov=hfov, vfov=vfov, normalize_z=True)
    rays2d = np.reshape(np.transpose(rays, axes=[2, 0, 1]), [3, -1])
    return {
        "hfov": hfov, "vfov": vfov,
        "unit_ray_array": rays,
        "unit_ray_array_2D": torch.from_numpy(rays2d.astype(np.float32)),
        "intrinsic_M_cuda": torch.from_numpy(K.astype(np.float32)),
        "intrinsic_M": K,
    }
