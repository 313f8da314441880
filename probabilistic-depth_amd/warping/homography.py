"""Drop-in for the reference's warping/homography.py hot-path functions.

Same names, positional order, argument meaning and error behaviour as the reference
(est_swp_volume_v4: warping/homography.py:98-135; warp_feature: :137-168); the work is done
by the HIP kernels behind include/pdepth.h.  Inputs must live on the GPU -- there is no CPU
path here (the CPU restatement lives in oracle/ and is test infrastructure only).
"""
import numpy as np
import torch

from .. import ops


def _camera_tensors(cam_intrinsic, device):
    K = cam_intrinsic["intrinsic_M_cuda"].to(device=device, dtype=torch.float32).reshape(1, 3, 3)
    rays = cam_intrinsic["unit_ray_array_2D"].to(device=device, dtype=torch.float32)
    rays = rays.reshape(1, 3, -1)
    M = cam_intrinsic["intrinsic_M"]
    if isinstance(M, torch.Tensor):
        cxcy = M[:2, 2].to(device=device, dtype=torch.float32).reshape(1, 2)
    else:  # numpy copy, as built at models/models.py:538 -- u_center/v_center of homography.py:194
        cxcy = torch.tensor([[np.float32(M[0, 2]), np.float32(M[1, 2])]], dtype=torch.float32, device=device)
    return K, rays, cxcy


def est_swp_volume_v4(feat_img_ref, feat_img_src, d_candi, R, t, cam_intrinsic, costV_sigma,
                      feat_dist="L2", debug_ipdb=False, blas=None):
    r"""
    feat_img_ref - NCHW tensor (N == 1)
    feat_img_src - NVCHW tensor.  V is for different views
    R, t - R[idx_view, :, :] - 3x3 rotation matrix
           t[idx_view, :] - 3x1 transition vector
    returns costV [1, D, H, W] on feat_img_ref.device (new tensor, inputs untouched)
    """
    device = feat_img_ref.device
    V = feat_img_src.shape[1]
    K, rays, cxcy = _camera_tensors(cam_intrinsic, device)
    R = R.to(device=device, dtype=torch.float32).reshape(1, V, 3, 3)
    t = t.to(device=device, dtype=torch.float32).reshape(1, V, 3)
    return ops.sweep_cost(feat_img_ref[:1], feat_img_src[:1], K, R, t, rays, cxcy, d_candi, costV_sigma,
                          feat_dist=feat_dist, blas=blas)


def warp_feature(feat_img_src, d_candi, R, t, cam_intrinsic, blas=None):
    r"""
    feat_img_src - NVCHW tensor (N == 1, C == len(d_candi)); channel i is warped with plane i
    returns [1, V, C, H, W]
    """
    if feat_img_src.shape[0] != 1:
        raise Exception("Warped Accum Error")
    device = feat_img_src.device
    V = feat_img_src.shape[1]
    K, rays, cxcy = _camera_tensors(cam_intrinsic, device)
    R = R.to(device=device, dtype=torch.float32).reshape(1, V, 3, 3)
    t = t.to(device=device, dtype=torch.float32).reshape(1, V, 3)
    return ops.warp_feature(feat_img_src, K, R, t, rays, cxcy, d_candi, blas=blas)


def get_rel_extrinsicM(ext_ref, ext_src):
    """Extrinsic matrix from ref view to src view (warping/homography.py:260-262; used by the loader)."""
    return ext_src.dot(np.linalg.inv(ext_ref))
