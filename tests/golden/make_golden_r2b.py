"""Golden fixture g17: autograd through the reference's inverse_warp (utils/inverse_warp.py:174-210), the way its two
callers use it (losses/loss_blocks.py:116 bilinear, :151 'nearest'): gradients with respect to the image, the depth map
and the 6-DoF pose, for a seeded upstream gradient.  Build container only (imports /root/reference):

    python tests/golden/make_golden_r2b.py
"""
import os
import sys
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _import_reference  # noqa: E402,F401  (sets sys.path for the reference)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    _import_reference()
    import utils.inverse_warp as iw
    warnings.simplefilter("ignore")
    g = torch.Generator().manual_seed(1717)
    B, C, H, W = 2, 3, 24, 36
    img = torch.randn(B, C, H, W, generator=g)
    depth = torch.rand(B, H, W, generator=g) * 20 + 4
    K = torch.tensor([[[38.0, 0, 18.3], [0, 35.0, 11.6], [0, 0, 1]]]).repeat(B, 1, 1)
    pose6 = torch.tensor([[0.25, -0.1, 0.3, 0.02, -0.03, 0.01], [-0.4, 0.05, 0.1, -0.01, 0.02, 0.04]])
    gout = torch.randn(B, C, H, W, generator=g)
    out = {}
    for mode in ("bilinear", "nearest"):
        for rot in ("euler", "quat"):
            i_, d_, p_ = img.clone().requires_grad_(True), depth.clone().requires_grad_(True), pose6.clone().requires_grad_(True)
            o, valid = iw.inverse_warp(i_, d_, p_, K, mode, rot)
            (o * gout).sum().backward()
            tag = mode + "_" + rot
            out[tag + "_out"] = o.detach().numpy()
            out[tag + "_valid"] = valid.numpy()
            out[tag + "_gimg"] = i_.grad.numpy()
            out[tag + "_gdepth"] = d_.grad.numpy() if d_.grad is not None else np.zeros_like(depth.numpy())
            out[tag + "_gpose"] = p_.grad.numpy() if p_.grad is not None else np.zeros_like(pose6.numpy())
    # a 4x4 pose and gradients to the intrinsics as well
    pose44 = torch.eye(4).repeat(B, 1, 1)
    pose44[:, :3, 3] = torch.tensor([[0.3, -0.1, 0.2], [-0.5, 0.0, 0.1]])
    p44, k_ = pose44.clone().requires_grad_(True), K.clone().requires_grad_(True)
    d_ = depth.clone().requires_grad_(True)
    o, _ = iw.inverse_warp(img, d_, p44, k_)
    (o * gout).sum().backward()
    np.savez_compressed(os.path.join(HERE, "g17_inverse_warp_backward.npz"), img=img.numpy(), depth=depth.numpy(), K=K.numpy(),
                        pose6=pose6.numpy(), pose44=pose44.numpy(), grad_out=gout.numpy(), p44_out=o.detach().numpy(),
                        p44_gdepth=d_.grad.numpy(), p44_gpose=p44.grad.numpy(), p44_gK=k_.grad.numpy(),
                        meta_torch=torch.__version__, **out)
    print("g17_inverse_warp_backward.npz", os.path.getsize(os.path.join(HERE, "g17_inverse_warp_backward.npz")), "bytes")
    for k_, v in out.items():
        if k_.endswith("gpose") or k_.endswith("gdepth"):
            print(k_, float(np.abs(v).max()))


if __name__ == "__main__":
    main()
