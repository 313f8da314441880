"""Round-2 golden fixtures (g13 .. g16), generated from the REAL reference like make_golden.py.

Run in the build container only (needs /root/reference, which never travels):

    python tests/golden/make_golden_r2.py

Nothing of the reference is copied into the repository: its functions are imported and called, and -- for the variance
lines that exist only inline in the evaluation loop (trainer/default_trainer.py:333-336) -- the four statements are read
from the reference file AT RUN TIME and executed on a seeded tensor (with `.cuda()` removed: this container has no GPU).
Fixtures are data only (inputs by value or by seed, outputs by value).
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _import_reference, _rays_and_K  # noqa: E402  (also sets sys.path for reference + repo)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def save(name, **arrays):
    np.savez_compressed(os.path.join(HERE, name), **arrays)


def peaked_logdpv(gen, D, H, W, d_candi, spread=1.5):
    """A smooth, peaked log-DPV: depth rising from the top to the bottom rows plus low-frequency noise."""
    yy = torch.linspace(0, 1, H)[:, None].expand(H, W)
    centre = 8.0 + 20.0 * yy + 3.0 * torch.sin(torch.linspace(0, 6.0, W))[None, :] + torch.randn(H, W, generator=gen) * 0.3
    d = torch.tensor(d_candi, dtype=torch.float32)[:, None, None]
    logits = -((d - centre[None]) ** 2) / (2 * spread ** 2) + 0.1 * torch.randn(D, H, W, generator=gen)
    return F.log_softmax(logits[None], dim=1)


def main():
    homo, view, img_utils = _import_reference()
    import pdepth_amd
    from pdepth_amd import _native, synth as S
    from util_host import cpu_vendor
    meta = dict(meta_torch=torch.__version__, meta_cpu_capability=torch.backends.cpu.get_cpu_capability(),
                meta_cpu_vendor=cpu_vendor(), meta_blas_mode=np.int32(_native.host_blas_mode()))

    # ---- G13: variance of the depth distribution, the evaluation loop's inline lines ------------------------
    lines = open(os.path.join(REF, "trainer", "default_trainer.py")).read().split("\n")[332:336]   # lines 333..336
    assert lines[0].strip().startswith("z = torch.exp(") and lines[3].strip().startswith("variance ="), lines
    g13 = torch.Generator().manual_seed(1313)
    d13 = img_utils.powerf(5.0, 40.0, 64, 1.0)
    dpv13 = peaked_logdpv(g13, 64, 24, 40, d13, spread=2.5)
    ns = {"torch": torch, "dpv_refined_predicted": dpv13, "d_candi": d13}
    exec("\n".join(ln.strip().replace(".cuda()", "") for ln in lines), ns)
    save("g13_variance.npz", logdpv=dpv13.numpy(), d_candi=d13, mean=ns["mean"].numpy(), variance=ns["variance"].numpy(),
         mean_dtype=str(ns["mean"].dtype), **meta)

    # ---- G14: uncertainty-field collapse (utils/img_utils.py:268-358) -----------------------------------------
    g14 = torch.Generator().manual_seed(1414)
    D, H, W = 32, 48, 64
    d14 = img_utils.powerf(5.0, 40.0, D, 1.0)
    intr = torch.tensor([[90.0, 0.0, 31.3], [0.0, 85.0, 18.6], [0.0, 0.0, 1.0]])
    # The reference calls Tensor.repeat([D, 1, 1], 0, 1) (utils/img_utils.py:342); torch 2.10 rejects the two stray
    # positional arguments older releases ignored.  For the duration of these calls repeat() drops them -- the
    # reference file itself is not touched.
    orig_repeat = torch.Tensor.repeat
    torch.Tensor.repeat = lambda self, *a: orig_repeat(self, a[0]) if a and isinstance(a[0], (list, tuple)) else orig_repeat(self, *a)
    out = {}
    for tag, cfgx in (("a", {"unc_ang": 5, "unc_shift": 0.6, "unc_span": 0.9}),      # shifted by 5 rows
                      ("b", {"unc_ang": 0, "unc_shift": 0.5, "unc_span": 1.2})):     # no shift
        logdpv = peaked_logdpv(g14, D, H, W, d14)
        mask = (torch.rand(1, H, W, generator=g14) > 0.3).float()
        # predicted volume: log-DPV, no mask; "truth" volume: probabilities + validity mask
        # (the two calls of compute_unc_field, utils/img_utils.py:178-181)
        pl, dz = img_utils.gen_ufield(logdpv, d14, intr, BV_log=True, cfgx=cfgx)
        pt, dzt = img_utils.gen_ufield(torch.exp(logdpv), d14, intr, BV_log=False, mask=mask, cfgx=cfgx)
        pn, _ = img_utils.gen_ufield(logdpv, d14, intr, BV_log=True, normalize=True, cfgx=cfgx)
        out.update({f"{tag}_logdpv": logdpv.numpy(), f"{tag}_mask": mask.numpy(), f"{tag}_cfgx": np.array(
            [cfgx["unc_ang"], cfgx["unc_shift"], cfgx["unc_span"]], dtype=np.float64),
            f"{tag}_plane_log": pl.numpy(), f"{tag}_depthzero_log": dz.numpy(), f"{tag}_plane_prob_masked": pt.numpy(),
            f"{tag}_depthzero_prob_masked": dzt.numpy(), f"{tag}_plane_log_normalized": pn.numpy()})
        frac = float((dz != 0).float().mean())
        assert 0.02 < frac < 0.9, f"degenerate mask in fixture {tag}: {frac}"
    torch.Tensor.repeat = orig_repeat
    save("g14_ufield.npz", d_candi=d14, intr=intr.numpy(), **out, **meta)

    # ---- G15: 5-frame feedback trajectory at a 256x512 image (BASELINE config 4) -------------------------------
    import models.get_model as gm
    torch.nn.Module.cuda = lambda self, *a, **k: self     # models.py:399 calls .cuda() on the residual blocks
    cfg = S.default_cfg("default_feedback")
    torch.manual_seed(0)
    model = gm.get_model(cfg, 0)
    S.seed_weights(model, seed=15)
    model.eval()
    prev, lows, refs = None, [], []
    with torch.no_grad():
        for frame in range(5):
            inp = S.make_model_input(15000 + frame, B=1, V=1, H=256, W=512, D=64, pose="mono")
            inp["prev_output"] = prev
            o = model([inp])[0]
            prev = F.interpolate(o["output_refined"][-1].detach(), scale_factor=0.25, mode="nearest")  # default_trainer.py:221
            lows.append(img_utils.dpv_to_depthmap(o["output"][-1][0:1], inp["d_candi"], BV_log=True).numpy())
            refs.append(img_utils.dpv_to_depthmap(o["output_refined"][-1][0:1], inp["d_candi"], BV_log=True).numpy()[:, ::4, ::4])
            print("frame", frame, "depth range", lows[-1].min(), lows[-1].max())
    save("g15_trajectory.npz", depth_low=np.concatenate(lows), depth_ref_sub=np.concatenate(refs), seed=np.int32(15),
         first_input_seed=np.int32(15000), **meta)

    # ---- G16: the PackNet-style head (models/packnet.py:380-394): sweep -> log_softmax -> expectation ---------
    it = S.make_item(16000, C=67, D=64, H=64, W=96, V=1, pose="stereo")
    K64 = it["K"].numpy().astype(np.float64)
    cam = {"intrinsic_M_cuda": it["K"], "intrinsic_M": it["K"].numpy(), "unit_ray_array_2D": it["rays"]}
    costv = homo.est_swp_volume_v4(it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], cam, 10.0, feat_dist="L2")
    bv = F.log_softmax(costv, dim=1)                                   # packnet.py:394
    dep = img_utils.dpv_to_depthmap(bv, it["d_candi"], BV_log=True)
    save("g16_packnet_head.npz", seed=np.int32(16000), logdpv_sub=bv.numpy()[:, ::4, ::2, ::2], depth=dep.numpy(), **meta)

    for f in sorted(os.listdir(HERE)):
        if f.startswith(("g13", "g14", "g15", "g16")):
            print("  %-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))


if __name__ == "__main__":
    main()
