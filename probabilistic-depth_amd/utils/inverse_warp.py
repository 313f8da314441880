"""Drop-in for the reference's utils/inverse_warp.py::inverse_warp (forward only).

Same signature and return values as utils/inverse_warp.py:174-210.  The 3x3 / 3x4 camera algebra
(intrinsics.inverse(), pose vector -> matrix :140-156 with euler2mat :72-108 / quat2mat :110-131,
intrinsics @ pose :200) stays in torch on the device -- a few dozen flops per batch item -- and the per-pixel
back-projection, projection and bilinear sampling run in one HIP kernel (pdepth_inverse_warp_f32).
The reference uses this function only in its training losses (losses/loss_blocks.py:116,151); there is no
backward here, so tensors that require grad raise.
"""
import torch

from .. import _native


def euler2mat(angle):
    """[B,3] rotation angles about x, y, z (radians) -> [B,3,3] = Rx @ Ry @ Rz (utils/inverse_warp.py:72-108)."""
    x, y, z = angle[:, 0], angle[:, 1], angle[:, 2]
    zero, one = torch.zeros_like(z), torch.ones_like(z)
    cz, sz, cy, sy, cx, sx = torch.cos(z), torch.sin(z), torch.cos(y), torch.sin(y), torch.cos(x), torch.sin(x)
    zmat = torch.stack([cz, -sz, zero, sz, cz, zero, zero, zero, one], dim=1).reshape(-1, 3, 3)
    ymat = torch.stack([cy, zero, sy, zero, one, zero, -sy, zero, cy], dim=1).reshape(-1, 3, 3)
    xmat = torch.stack([one, zero, zero, zero, cx, -sx, zero, sx, cx], dim=1).reshape(-1, 3, 3)
    return torch.matmul(torch.matmul(xmat, ymat), zmat)


def quat2mat(quat):
    """[B,3] vector part of a quaternion with w = 1 before normalisation -> [B,3,3] (utils/inverse_warp.py:110-131)."""
    q = torch.cat([torch.ones_like(quat[:, :1]), quat], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).reshape(-1, 3, 3)


def pose_vec2mat(vec, rotation_mode="euler"):
    """[B,6] (tx,ty,tz,rx,ry,rz) -> [B,3,4] (utils/inverse_warp.py:134-156)."""
    rot = euler2mat(vec[:, 3:]) if rotation_mode == "euler" else quat2mat(vec[:, 3:])
    return torch.cat([rot, vec[:, :3].unsqueeze(-1)], dim=2)


def inverse_warp(img, depth, pose, intrinsics, mode="bilinear", rotation_mode="euler", padding_mode="zeros"):
    """Inverse warp a source image to the target image plane -> (projected_img [B,C,H,W], valid_points bool [B,H,W])."""
    if depth.dim() != 3:
        raise AssertionError("wrong size for depth, expected BxHxW, got  {}".format(list(depth.size())))
    if intrinsics.dim() != 3 or intrinsics.shape[1:] != (3, 3):
        raise AssertionError("wrong size for intrinsics, expected Bx3x3, got  {}".format(list(intrinsics.size())))
    if mode != "bilinear" or padding_mode != "zeros":
        raise NotImplementedError("inverse_warp: the HIP path implements mode='bilinear', padding_mode='zeros'")
    if img.requires_grad or depth.requires_grad or pose.requires_grad:
        raise RuntimeError("inverse_warp: backward is not implemented in the HIP path (forward/eval only)")
    if pose.shape[1] == 6:
        pose_mat = pose_vec2mat(pose, rotation_mode)
    elif pose.shape[1] == 4 and pose.shape[2] == 4:
        pose_mat = pose[:, 0:3, :]
    else:
        raise RuntimeError("inverse_warp: pose must be [B,6] or [B,4,4]")
    intrinsics = intrinsics.float()
    proj = torch.matmul(intrinsics, pose_mat.float())
    return _native.inverse_warp(img.float(), depth.float(), intrinsics.inverse(), proj)
