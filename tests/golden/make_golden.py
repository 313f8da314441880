"""Generate the golden fixtures in this directory from the REAL reference.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden.py

It imports the reference's own hot-path functions (warping/homography.py, warping/view.py,
utils/img_utils.py) -- nothing is copied -- evaluates them on small seeded inputs on the CPU
and stores inputs + outputs as .npz.  The fixtures pin oracle/ref_cpu.py (tests -m "not gpu")
and the HIP kernels (tests -m gpu).  Fixtures are data only.
"""
import os
import sys
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def _import_reference():
    """Reference modules; cv2 / torchvision / deval_lib are absent here and unused by the
    functions we call, so inert placeholders satisfy the import statements."""
    for name in ("cv2", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    import external.deval_lib as dl  # real (empty) package of the reference
    stub = types.ModuleType("external.deval_lib.pyevaluatedepth_lib")
    sys.modules["external.deval_lib.pyevaluatedepth_lib"] = stub
    dl.pyevaluatedepth_lib = stub
    import warping.homography as homo
    import warping.view as view
    import utils.img_utils as img_utils
    return homo, view, img_utils


def _cam_dict(K64, rays):
    K32 = torch.from_numpy(K64.astype(np.float32))
    return {"intrinsic_M_cuda": K32, "intrinsic_M": K32.cpu().numpy(), "unit_ray_array_2D": rays}


def _rot(yaw, pitch, roll=0.0):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    Rz = np.array([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
    return Ry @ Rx @ Rz


def _rays_and_K(view, w, h, hfov, vfov):
    import math
    rays = view.normalised_pixel_to_ray_array(width=w, height=h, hfov=hfov, vfov=vfov, normalize_z=True)
    rays2d = torch.from_numpy(np.reshape(np.transpose(rays, axes=[2, 0, 1]), [3, -1]).astype(np.float32))
    K = np.zeros((3, 3))
    K[2, 2] = 1.0  # kittiloader/kitti.py:284-293
    K[0, 0] = (w / 2.0) / math.tan(math.radians(hfov / 2.0))
    K[0, 2] = w / 2.0
    K[1, 1] = (h / 2.0) / math.tan(math.radians(vfov / 2.0))
    K[1, 2] = h / 2.0
    return rays2d, K


def main():
    homo, view, img_utils = _import_reference()
    torch.manual_seed(1234)
    import pdepth_amd
    from util_host import cpu_vendor
    # blas_mode: how THIS host's MKL rounds K@R / K@t / (K@R)@rays (include/pdepth.h PDEPTH_BLAS_*);
    # the HIP kernels are asked for the same mode when they are checked against these fixtures.
    meta = dict(torch=torch.__version__, cpu_capability=torch.backends.cpu.get_cpu_capability(),
                cpu_vendor=cpu_vendor(), blas_mode=np.int32(pdepth_amd._native.host_blas_mode()))
    only = os.environ.get("PDEPTH_GOLDEN_ONLY")  # e.g. "g12": rewrite just that fixture, leave the others untouched

    def save(name, **kw):
        if only and not name.startswith(only):
            return
        np.savez_compressed(os.path.join(HERE, name), **kw, **{"meta_" + k: v for k, v in meta.items()})

    # ---- G1/G2/G3: tiny sweeps (16x24, C=7, D=8, V=2), off-centre principal point -------
    h, w, C, D, V = 16, 24, 7, 8, 2
    rays, K = _rays_and_K(view, w, h, 80.0, 35.0)
    K[0, 2] += 1.3
    K[1, 2] -= 0.7
    cam = _cam_dict(K, rays)
    d_candi = img_utils.powerf(5.0, 40.0, D, 1.0)
    ref = torch.randn(1, C, h, w)
    src = torch.randn(1, V, C, h, w)
    poses = {
        "g1_rot_trans": (np.stack([_rot(0.02, -0.01, 0.005), _rot(-0.03, 0.0)]), np.array([[0.3, 0.02, 0.1], [-0.54, 0.0, 0.0]])),
        "g2_identity": (np.stack([np.eye(3), np.eye(3)]), np.zeros((2, 3))),
        "g3_out_of_bounds": (np.stack([np.eye(3), _rot(0.4, 0.0)]), np.array([[30.0, 0.0, 0.0], [0.0, -9.0, 0.5]])),
    }
    for name, (Rn, tn) in poses.items():
        R = torch.from_numpy(Rn.astype(np.float32))
        t = torch.from_numpy(tn.astype(np.float32))
        out = {}
        for metric in ("L2", "L1"):
            out["cost_" + metric] = homo.est_swp_volume_v4(ref, src, d_candi, R, t, cam, 10.0, feat_dist=metric).numpy()
        save(name + ".npz", ref=ref.numpy(), src=src.numpy(), K=cam["intrinsic_M_cuda"].numpy(), R=R.numpy(),
             t=t.numpy(), rays=rays.numpy(), d_candi=d_candi, sigma=np.float32(10.0), **out)

    # ---- G4: model-real sizes, inputs regenerated from the seed by the package's synth ---
    import pdepth_amd  # noqa: F401  (alias of probabilistic-depth_amd)
    from pdepth_amd import synth
    for name, kw in (("g4_stereo_64x96", dict(seed=4001, C=67, D=64, H=64, W=96, V=1, pose="stereo")),
                     ("g4_mono_64x128", dict(seed=4002, C=67, D=64, H=64, W=128, V=1, pose="mono"))):
        it = synth.make_item(**kw)
        cam4 = {"intrinsic_M_cuda": it["K"], "intrinsic_M": it["K"].numpy(), "unit_ray_array_2D": it["rays"]}
        cost = homo.est_swp_volume_v4(it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], cam4, 10.0)
        logp = F.log_softmax(cost, dim=1)  # models/packnet.py:394
        depth = img_utils.dpv_to_depthmap(logp, it["d_candi"], BV_log=True)
        save(name + ".npz", synth_kwargs=np.array(repr(kw)), cost_sub=cost.numpy()[:, ::4, ::2, ::2],
             cost_sum=cost.double().sum().numpy(), depth=depth.numpy(), logp_sub=logp.numpy()[:, ::4, ::2, ::2])

    # ---- G5: DPV reduction ----------------------------------------------------------------
    D5, h5, w5 = 64, 32, 48
    d5 = img_utils.powerf(5.0, 40.0, D5, 1.0)
    broad = torch.randn(1, D5, h5, w5) * 0.3
    peaked = torch.randn(1, D5, h5, w5) * 6.0
    out = {}
    for nm, x in (("broad", broad), ("peaked", peaked)):
        lp = F.log_softmax(x, dim=1)  # models/models.py:560
        out[nm + "_logits"] = x.numpy()
        out[nm + "_logp"] = lp.numpy()
        out[nm + "_depth_log"] = img_utils.dpv_to_depthmap(lp, d5, BV_log=True).numpy()
        out[nm + "_depth_lin"] = img_utils.dpv_to_depthmap(torch.exp(lp), d5, BV_log=False).numpy()
    save("g5_dpv.npz", d_candi=d5, **out)

    # ---- G6: warp_feature (V=2 incl. identity view, C=D=8) ---------------------------------
    D6 = 8
    d6 = img_utils.powerf(5.0, 40.0, D6, 1.0)
    feat = torch.randn(1, 2, D6, h, w)
    R6 = torch.from_numpy(np.stack([_rot(0.02, -0.01), np.eye(3)]).astype(np.float32))
    t6 = torch.from_numpy(np.array([[0.3, 0.02, 0.1], [0, 0, 0]], dtype=np.float32))
    wf = homo.warp_feature(feat, d6, R6, t6, cam)
    save("g6_warp_feature.npz", feat=feat.numpy(), K=cam["intrinsic_M_cuda"].numpy(), R=R6.numpy(), t=t6.numpy(),
         rays=rays.numpy(), d_candi=d6, out=wf.numpy())

    # ---- G7: host-side producers ------------------------------------------------------------
    g7 = {}
    for p in (1.0, 1.25, 1.5):
        g7["powerf_%g" % p] = img_utils.powerf(5.0, 40.0, 64, p)
    for (ww, hh) in ((96, 64), (128, 64)):
        r, Kk = _rays_and_K(view, ww, hh, 80.0, 35.0)
        g7["rays_%dx%d" % (ww, hh)] = r.numpy()
        g7["K_%dx%d" % (ww, hh)] = Kk
    save("g7_host.npz", **g7)
    # ---- G8: whole-model captures (reference BaseModel on the CPU, name-seeded weights) ----------
    import models.models as ref_models  # noqa: F401
    import models.get_model as ref_get_model
    from pdepth_amd import synth as S
    torch.nn.Module.cuda = lambda self, *a, **k: self  # reference Base3D calls .cuda(id) in its ctor
    g8 = {}
    for nmode in ("default", "default_feedback"):
        cfg = S.default_cfg(nmode)
        torch.manual_seed(0)
        model = ref_get_model.get_model(cfg, 0)
        S.seed_weights(model, seed=8)
        model.eval()
        keys = list(model.state_dict().keys())
        g8[nmode + "_state_keys"] = np.array(keys)
        g8[nmode + "_state_shapes"] = np.array([repr(tuple(v.shape)) for v in model.state_dict().values()])
        prev = None
        for frame in range(2 if nmode == "default_feedback" else 1):
            inp = S.make_model_input(8000 + frame, B=1, V=1, H=256, W=256, D=64, pose="mono")
            inp["prev_output"] = prev
            with torch.no_grad():
                out = model([inp])[0]
                if nmode == "default" and frame == 0:
                    _, costv, _, _ = model.forward_encoder(inp)
                    g8["default_cost_sub"] = costv.numpy()[:, ::4, ::2, ::2]
            tag = "%s_f%d" % (nmode, frame)
            g8[tag + "_logdpv_sub"] = out["output"][-1].numpy()[:, ::4, ::2, ::2]
            g8[tag + "_depth_low"] = img_utils.dpv_to_depthmap(out["output"][-1], inp["d_candi"], BV_log=True).numpy()
            g8[tag + "_depth_ref"] = img_utils.dpv_to_depthmap(out["output_refined"][-1], inp["d_candi"], BV_log=True).numpy()
            prev = F.interpolate(out["output_refined"][-1].detach(), scale_factor=0.25, mode="nearest")
    save("g8_model.npz", **g8)

    # ---- G9: correlation (reference's pure-PyTorch twin of its CUDA op) ---------------------------
    import models.correlation_native as corr_native
    x1, x2 = torch.randn(2, 32, 12, 16), torch.randn(2, 32, 12, 16)
    save("g9_correlation.npz", x1=x1.numpy(), x2=x2.numpy(), out=corr_native.Correlation(max_displacement=4)(x1, x2).numpy())

    # ---- G10: DPV Bayesian fusion of the upsample mode (models/models.py:663-672) -----------------
    D10, h10, w10 = 64, 24, 40
    d10 = img_utils.powerf(5.0, 40.0, D10, 1.0)
    bv = F.log_softmax(torch.randn(2, D10, h10, w10) * 2.0, dim=1)
    dm = torch.rand(2, h10, w10) * 30.0 + 6.0
    mk = (torch.rand(2, 1, h10, w10) > 0.6).float()
    dm = dm * mk[:, 0]
    tofuse = img_utils.gen_dpv_withmask(dm, mk, d10, 0.3)
    fused = torch.exp(bv + torch.log(tofuse))
    fused = fused / torch.sum(fused, dim=1).unsqueeze(1)
    fused = torch.clamp(fused, img_utils.epsilon, 1.)
    save("g10_dpv_fuse.npz", logp=bv.numpy(), dmaps=dm.numpy(), masks=mk.numpy(), d_candi=d10, tofuse=tofuse.numpy(),
         fused=fused.numpy(), logfused=torch.log(fused).numpy())

    # ---- G11: inverse_warp (training-loss warp named by the north star) ----------------------------
    import utils.inverse_warp as iw
    hh, ww = 20, 28
    img11 = torch.randn(2, 3, hh, ww)
    dep11 = torch.rand(2, hh, ww) * 20 + 4
    K11 = torch.tensor([[[30.0, 0, 14.2], [0, 28.0, 9.7], [0, 0, 1]]]).repeat(2, 1, 1)
    pose44 = torch.eye(4).repeat(2, 1, 1)
    pose44[0, :3, :3] = torch.from_numpy(_rot(0.03, -0.02, 0.01).astype(np.float32)); pose44[0, :3, 3] = torch.tensor([0.4, -0.1, 0.3])
    pose44[1, :3, 3] = torch.tensor([-0.6, 0.05, -0.2])
    pose6 = torch.tensor([[0.2, -0.1, 0.3, 0.02, -0.03, 0.01], [-0.4, 0.0, 0.1, -0.01, 0.02, 0.04]])
    o44, v44 = iw.inverse_warp(img11, dep11, pose44, K11)
    o6e, v6e = iw.inverse_warp(img11, dep11, pose6, K11, rotation_mode="euler")
    o6q, v6q = iw.inverse_warp(img11, dep11, pose6, K11, rotation_mode="quat")
    save("g11_inverse_warp.npz", img=img11.numpy(), depth=dep11.numpy(), K=K11.numpy(), pose44=pose44.numpy(),
         pose6=pose6.numpy(), out44=o44.numpy(), valid44=v44.numpy(), out6e=o6e.numpy(), valid6e=v6e.numpy(),
         out6q=o6q.numpy(), valid6q=v6q.numpy())

    # ---- G12: correlation backward: autograd through the reference's pure-PyTorch twin -------------
    g12 = torch.Generator().manual_seed(1212)
    x1 = torch.randn(2, 20, 11, 14, generator=g12).requires_grad_(True)
    x2 = torch.randn(2, 20, 11, 14, generator=g12).requires_grad_(True)
    go = torch.randn(2, 81, 11, 14, generator=g12)
    out12 = corr_native.Correlation(max_displacement=4)(x1, x2)
    out12.backward(go)
    save("g12_correlation_backward.npz", x1=x1.detach().numpy(), x2=x2.detach().numpy(), grad_out=go.numpy(),
         grad_x1=x1.grad.numpy(), grad_x2=x2.grad.numpy())

    print("golden fixtures written to", HERE)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("  %-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
