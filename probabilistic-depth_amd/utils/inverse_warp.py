"""Drop-in for the reference's utils/inverse_warp.py::inverse_warp, forward and backward.

Same signature and return values as utils/inverse_warp.py:174-210.  The 3x3 / 3x4 camera algebra
(intrinsics.inverse(), pose vector -> matrix :140-156 with euler2mat :72-108 / quat2mat :110-131,
intrinsics @ pose :200) stays in torch on the device -- a few dozen flops per batch item, differentiable by torch's
own autograd -- and the per-pixel back-projection, projection and sampling run in one HIP kernel
(pdepth_inverse_warp_f32; 'bilinear' and 'nearest', zeros padding).  The reference uses this function in its training
losses, under autograd (losses/loss_blocks.py:116,151): when an input requires grad the call goes through
_InverseWarpFn, whose backward is pdepth_inverse_warp_backward_f32 (image gradient by atomic scatter, gradient of the
projected point per pixel) followed by the small products that carry it to the depth map, the pose and the intrinsics.
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


class _InverseWarpFn(torch.autograd.Function):
    """(img, depth, Kinv, proj) -> (warped, valid): the HIP kernels with the chain rule of pixel2cam / cam2pixel."""

    @staticmethod
    def forward(ctx, img, depth, Kinv, proj, mode):
        out, valid = _native.inverse_warp(img, depth, Kinv, proj, mode)
        ctx.save_for_backward(img, depth, Kinv, proj)
        ctx.mode = mode
        ctx.mark_non_differentiable(valid)
        return out, valid

    @staticmethod
    def backward(ctx, g_out, _g_valid):
        img, depth, Kinv, proj = ctx.saved_tensors
        need_img = ctx.needs_input_grad[0]
        need_pt = any(ctx.needs_input_grad[1:4]) and ctx.mode == "bilinear"
        g_img = g_depth = g_Kinv = g_proj = None
        if need_img or need_pt:
            g_img, g_pc = _native.inverse_warp_backward(img, depth, Kinv, proj, g_out.float(), ctx.mode,
                                                        want_img=need_img, want_point=need_pt)
        if need_pt:
            B, _, H, W = img.shape
            ys, xs = torch.meshgrid(torch.arange(H, device=img.device, dtype=torch.float32),
                                    torch.arange(W, device=img.device, dtype=torch.float32), indexing="ij")
            pix = torch.stack([xs, ys, torch.ones_like(xs)]).reshape(1, 3, H * W)          # (x, y, 1) per pixel
            g_pc = g_pc.reshape(B, 3, H * W)
            d = depth.reshape(B, 1, H * W)
            cam0 = torch.matmul(Kinv, pix)                                                 # pixel2cam before the depth
            g_cam = torch.matmul(proj[:, :, :3].transpose(1, 2), g_pc)
            if ctx.needs_input_grad[1]:
                g_depth = (g_cam * cam0).sum(1).reshape(B, H, W)
            if ctx.needs_input_grad[2]:
                g_Kinv = torch.matmul(g_cam * d, pix.transpose(1, 2).expand(B, -1, -1))
            if ctx.needs_input_grad[3]:
                g_proj = torch.cat([torch.matmul(g_pc, (cam0 * d).transpose(1, 2)), g_pc.sum(2, keepdim=True)], dim=2)
        elif ctx.mode == "nearest":   # the output does not depend on the sample position
            g_depth = torch.zeros_like(depth) if ctx.needs_input_grad[1] else None
            g_Kinv = torch.zeros_like(Kinv) if ctx.needs_input_grad[2] else None
            g_proj = torch.zeros_like(proj) if ctx.needs_input_grad[3] else None
        return g_img, g_depth, g_Kinv, g_proj, None


def inverse_warp(img, depth, pose, intrinsics, mode="bilinear", rotation_mode="euler", padding_mode="zeros"):
    """Inverse warp a source image to the target image plane -> (projected_img [B,C,H,W], valid_points bool [B,H,W])."""
    if depth.dim() != 3:
        raise AssertionError("wrong size for depth, expected BxHxW, got  {}".format(list(depth.size())))
    if intrinsics.dim() != 3 or intrinsics.shape[1:] != (3, 3):
        raise AssertionError("wrong size for intrinsics, expected Bx3x3, got  {}".format(list(intrinsics.size())))
    if mode not in ("bilinear", "nearest") or padding_mode != "zeros":
        raise NotImplementedError("inverse_warp: the HIP path implements mode='bilinear' | 'nearest', padding_mode='zeros'")
    if pose.shape[1] == 6:
        pose_mat = pose_vec2mat(pose, rotation_mode)
    elif pose.shape[1] == 4 and pose.shape[2] == 4:
        pose_mat = pose[:, 0:3, :]
    else:
        raise RuntimeError("inverse_warp: pose must be [B,6] or [B,4,4]")
    intrinsics = intrinsics.float()
    proj = torch.matmul(intrinsics, pose_mat.float())
    Kinv = intrinsics.inverse()
    if torch.is_grad_enabled() and any(t.requires_grad for t in (img, depth, Kinv, proj)):
        return _InverseWarpFn.apply(img.float(), depth.float(), Kinv, proj, mode)
    return _native.inverse_warp(img.float(), depth.float(), Kinv, proj, mode)
