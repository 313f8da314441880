"""Round 3: the HIP ops of ONE feedback step on the reference's own tensors (fixture g18, tests/golden/make_golden_r3.py).

The end-to-end feedback comparison (test_model_host / g15) carries 2-3e-4 of depth through the 3-D convolutions (MIOpen on
the GPU against mkldnn where the fixture was made).  Here every op of the step that this package implements gets the
REFERENCE's inputs -- the convolutions' outputs included -- so what is asserted is the op, not the network:

  warp_feature                      models/models.py:616-625     1e-5 abs (bit-faithful positions, bilinear taps)
  log_softmax(BV_cur + BV_resi)     models/models.py:694         2e-5 abs on the log-DPV, 1e-4 m on E[d]
  the decoder's log_softmax + E[d]  models/models.py:351, trainer/default_trainer.py:230-233   the same bounds
CPU part: the oracle restatement reproduces the same tensors (pins the oracle on in-model data)."""
import numpy as np
import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import ops, synth
from oracle import ref_cpu as O
from util import golden, golden_blas

DEV = "cuda:0"
LOGP_ATOL, DEPTH_ATOL = 2e-5, 1e-4


def _frame(g):
    hw = [int(v) for v in g["image_hw"]]
    return synth.make_model_input(int(g["input_seed"]), B=1, V=1, H=hw[0], W=hw[1], D=64, pose="mono")


def test_oracle_reproduces_the_feedback_step_tensors():
    g = golden("g18_feedback_step.npz")
    inp = _frame(g)
    upd = torch.log_softmax(torch.from_numpy(g["BV_cur"]) + torch.from_numpy(g["BV_resi"]), dim=1)
    np.testing.assert_allclose(upd.numpy()[:, ::2], g["BV_upd_even"], rtol=0, atol=1e-6)
    depth = O.dpv_to_depthmap(upd, inp["d_candi"], BV_log=True)
    np.testing.assert_allclose(depth.numpy(), g["depth_low"], rtol=0, atol=2e-5)
    dec = torch.log_softmax(torch.from_numpy(g["dec_pre_crop"]), dim=1)
    np.testing.assert_allclose(dec.numpy(), g["dec_logp_crop"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.dpv_to_depthmap(dec, inp["d_candi"], BV_log=True).numpy(), g["depth_ref_crop"], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_hip_feedback_update_on_reference_tensors():
    """reduce_ex(addend): log_softmax(BV_cur + BV_resi), its exp (the decoder's input) and E[d] in one pass."""
    g = golden("g18_feedback_step.npz")
    inp = _frame(g)
    cur, resi = torch.from_numpy(g["BV_cur"]).to(DEV), torch.from_numpy(g["BV_resi"]).to(DEV)
    r = ops.dpv_reduce_ex(cur, inp["d_candi"], addend=resi, want_logp=True, want_prob=True, want_depth=True)
    logp = r["logp"].cpu().numpy()
    err_l = np.abs(logp[:, ::2] - g["BV_upd_even"]).max()
    err_d = np.abs(r["depth"].cpu().numpy() - g["depth_low"]).max()
    assert err_l <= LOGP_ATOL, f"log-DPV of the feedback update differs by {err_l:.2e}"
    assert err_d <= DEPTH_ATOL, f"E[d] of the feedback update differs by {err_d:.2e} m"
    np.testing.assert_allclose(r["prob"].cpu().numpy()[:, ::2], np.exp(g["BV_upd_even"]), rtol=2e-5, atol=1e-7)
    print(f"feedback update on reference tensors: logp {err_l:.2e}, depth {err_d:.2e} m")


@pytest.mark.gpu
def test_hip_decoder_dpv_pass_on_reference_tensors():
    """The decoder's last step (log_softmax over D at full resolution) + dpv_to_depthmap, on the reference's pre-softmax crop."""
    g = golden("g18_feedback_step.npz")
    inp = _frame(g)
    pre = torch.from_numpy(g["dec_pre_crop"]).to(DEV).contiguous()
    logp, depth = ops.dpv_reduce(pre, inp["d_candi"])
    err_l = np.abs(logp.cpu().numpy() - g["dec_logp_crop"]).max()
    err_d = np.abs(depth.cpu().numpy() - g["depth_ref_crop"]).max()
    assert err_l <= LOGP_ATOL and err_d <= DEPTH_ATOL, (err_l, err_d)
    r = ops.dpv_reduce_ex(pre, inp["d_candi"], want_logp=True, want_depth=True, want_quarter=True)
    assert np.abs(r["logp"].cpu().numpy() - g["dec_logp_crop"]).max() <= LOGP_ATOL
    assert np.array_equal(r["quarter"].cpu().numpy(), r["logp"].cpu().numpy()[:, :, ::4, ::4])   # default_trainer.py:221
    print(f"decoder DPV pass on reference tensors: logp {err_l:.2e}, depth {err_d:.2e} m")


@pytest.mark.gpu
def test_hip_warp_feature_on_reference_tensors():
    """The diagonal warp of the raw features (both views: the source view and the identity-posed reference view)."""
    g = golden("g18_feedback_step.npz")
    inp = _frame(g)
    feat = torch.from_numpy(g["feat_raw"]).to(DEV)            # [1, V+1, 64, h, w]
    poses = inp["src_cam_poses"].to(DEV)
    K = inp["intrinsics"].to(DEV)
    out = ops.warp_feature(feat, K, poses[:, :, :3, :3].contiguous(), poses[:, :, :3, 3].contiguous(), inp["unit_ray"].to(DEV),
                           K[:, :2, 2].contiguous(), inp["d_candi"], blas=golden_blas(g))
    err = np.abs(out.cpu().numpy()[:, :, ::2] - g["warped_even"]).max()
    scale = float(np.abs(g["warped_even"]).max())
    assert err <= 1e-5 * max(1.0, scale), f"warp_feature differs by {err:.2e} (values up to {scale:.2f})"
    print(f"warp_feature on reference tensors: {err:.2e} (values up to {scale:.2f})")


# ---------------------------------------------------------------------------------------------------------------
# the correlation op beyond the one configuration the reference instantiates
# ---------------------------------------------------------------------------------------------------------------
CORR_CONFIGS = [   # pad, kernel, max_displacement, stride1, stride2
    (4, 1, 4, 1, 1),      # the reference's own (fast path)
    (5, 1, 4, 1, 1),      # pad > max_displacement: the output grows by 2 * (pad - max_displacement)
    (3, 1, 4, 1, 2),      # pad < max_displacement: it shrinks
    (4, 1, 4, 2, 1),      # stride1 = 2
    (4, 3, 4, 1, 3),      # 3x3 kernel (max_displacement mod stride2 = 1 >= kernel radius)
    (6, 3, 5, 2, 2),      # everything at once
    (20, 1, 20, 1, 2),    # 21 x 21 displacements (FlowNet-C's head): beyond the fast path's radius
]


def test_oracle_general_correlation_contains_the_native_one():
    """The general restatement (from the kernel's index arithmetic) against the restatement of correlation_native.py that
    fixtures g9 / g12 pin: bit-identical where both apply."""
    g = torch.Generator().manual_seed(5)
    x1, x2 = torch.randn(2, 6, 13, 18, generator=g), torch.randn(2, 6, 13, 18, generator=g)
    for md in (1, 3, 4):
        assert torch.equal(O.correlation_general(x1, x2, md, 1, md, 1, 1), O.correlation(x1, x2, md))
    assert O.correlation_general(x1, x2, 6, 3, 5, 2, 2).shape == (2, 25, 7, 9)   # ceil((13 + 12 - 12) / 2), ceil((18 + 12 - 12) / 2)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", CORR_CONFIGS)
def test_hip_correlation_every_configuration(cfg):
    """Forward and backward (through ops.correlation's autograd Function, like correlation_package/correlation.py:6-44) for
    configurations of correlation_cuda_kernel.cu:41-114 that the reference never instantiates, in fp32 and with fp16 I/O."""
    pad, k, md, s1, s2 = cfg
    g = torch.Generator().manual_seed(100 + pad + 7 * k + 13 * s1)
    B, C, H, W = 2, 10, 21 if md < 10 else 44, 30 if md < 10 else 47
    x1, x2 = torch.randn(B, C, H, W, generator=g), torch.randn(B, C, H, W, generator=g)
    a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
    want = O.correlation_general(a, b, pad, k, md, s1, s2)
    go = torch.randn(want.shape, generator=g)
    want.backward(go)
    ad, bd = x1.to(DEV).requires_grad_(True), x2.to(DEV).requires_grad_(True)
    got = ops.correlation(ad, bd, pad, k, md, s1, s2)
    assert got.shape == want.shape
    np.testing.assert_allclose(got.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-5, atol=1e-6)
    got.backward(go.to(DEV))
    np.testing.assert_allclose(ad.grad.cpu().numpy(), a.grad.numpy(), rtol=1e-5, atol=5e-6)   # (sums of up to 441 x 9 terms)
    np.testing.assert_allclose(bd.grad.cpu().numpy(), b.grad.numpy(), rtol=1e-5, atol=5e-6)
    # fp16 tensors, fp32 accumulation (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, .cu:352-369):
    # against the oracle on the SAME fp16-rounded inputs, the only difference is the rounding of the outputs
    h1, h2 = x1.half(), x2.half()
    want16 = O.correlation_general(h1.float(), h2.float(), pad, k, md, s1, s2)
    got16 = ops.correlation(h1.to(DEV), h2.to(DEV), pad, k, md, s1, s2)
    assert got16.dtype == torch.float16
    np.testing.assert_allclose(got16.float().cpu().numpy(), want16.numpy(), rtol=1e-3, atol=1e-3)
    hd1, hd2 = h1.to(DEV).requires_grad_(True), h2.to(DEV).requires_grad_(True)
    ops.correlation(hd1, hd2, pad, k, md, s1, s2).backward(go.half().to(DEV))
    r1, r2 = h1.float().requires_grad_(True), h2.float().requires_grad_(True)
    O.correlation_general(r1, r2, pad, k, md, s1, s2).backward(go.half().float())
    np.testing.assert_allclose(hd1.grad.float().cpu().numpy(), r1.grad.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(hd2.grad.float().cpu().numpy(), r2.grad.numpy(), rtol=2e-3, atol=2e-3)


@pytest.mark.gpu
def test_hip_correlation_refuses_what_the_reference_kernel_reads_out_of_bounds_for():
    x = torch.randn(1, 4, 12, 12, device=DEV)
    for bad in ((4, 3, 4, 1, 1), (4, 2, 4, 1, 1), (4, 1, 4, 0, 1), (0, 1, 6, 1, 1)):   # radius > md mod s2; even kernel; stride 0; empty output
        with pytest.raises(RuntimeError):
            ops.correlation(x, x, *bad)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 128, 128), (128, 256, 64)])
def test_hip_correlation_reference_self_check(shape):
    """The reference's own numerical assertion (models/correlation_native.py:41-64): batch 4, C in {128, 256}, H in {128, 256},
    W in {64, 128}, N(0,1) inputs, max_displacement 4, the CUDA op equal to correlation_native at atol 1e-7.  Here: the HIP op
    against the restatement of correlation_native (bit-identical to the reference's on fixture g9), same shapes, same bound."""
    C, H, W = shape
    g = torch.Generator().manual_seed(C + H + W)
    x1, x2 = torch.randn(4, C, H, W, generator=g), torch.randn(4, C, H, W, generator=g)
    want = O.correlation(x1, x2, 4)
    got = ops.correlation(x1.to(DEV), x2.to(DEV), pad_size=4, kernel_size=1, max_displacement=4, stride1=1, stride2=1).cpu()
    assert torch.allclose(want, got, atol=1e-7), f"max |diff| {float((want - got).abs().max()):.2e}"


# ---------------------------------------------------------------------------------------------------------------
# the encoder epilogue kernel: cat + avg_pool2d + re-layout + Gram planes in one pass
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 2, 64, 64, 128, 4), (1, 3, 61, 37, 83, 4), (1, 2, 64, 64, 96, 2)])
def test_hip_pack_views_is_cat_plus_pack_source(shape):
    """ops.pack_views(feat, rgb) against the op sequence it replaces -- torch.cat((feat, F.avg_pool2d(rgb, rate))) (models.py:518-520),
    the view split (:530-534) and pdepth_pack_source_f32: same reference-view tensor, and a sweep on the packed views that equals
    the sweep on the concatenated tensor (the learned channels are copied bit for bit; the pooled image may differ from
    ATen's GPU kernel in the last place)."""
    import torch.nn.functional as F
    B, V1, Cf, h, w, rate = shape
    g = torch.Generator().manual_seed(sum(shape))
    feat = torch.randn(B * V1, Cf, h, w, generator=g).to(DEV)
    rgb = torch.rand(B * V1, 3, h * rate, w * rate, generator=g).to(DEV)
    packed, ref = ops.pack_views(feat, rgb, V1, 64)
    both = torch.cat((feat, F.avg_pool2d(rgb, rate)), dim=1).view(B, V1, Cf + 3, h, w)
    assert torch.equal(ref[:, :Cf], both[:, -1, :Cf])
    assert (ref[:, Cf:] - both[:, -1, Cf:]).abs().max().item() <= 1e-6
    b = synth.make_batch(77, B, C=Cf + 3, D=64, H=h, W=w, V=V1 - 1, pose="mono")
    d = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    args = (d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
    c_packed = ops.sweep_cost(ref, packed, *args)
    c_plain = ops.sweep_cost(both[:, -1].contiguous(), both[:, :-1].contiguous(), *args)
    rel = ((c_packed - c_plain).abs() / (1.0 + c_plain.abs())).max().item()
    assert rel < 2e-6, rel


@pytest.mark.gpu
@pytest.mark.parametrize("nmode", ["default", "default_feedback"])
def test_model_on_the_packed_epilogue_equals_the_concatenating_path(nmode):
    """BaseModel with the encoder epilogue kernel + packed sweep entry (default) against the same model with
    torch.cat + avg_pool2d + the plain entry: the same outputs to rounding."""
    from pdepth_amd.models import get_model
    torch.manual_seed(0)
    model = get_model(synth.default_cfg(nmode), 0).to(DEV).eval()
    synth.seed_weights(model, seed=21)
    inp = synth.make_model_input(21000, B=2, V=1, H=256, W=256, D=64, pose="mono")
    inp = {k: (v.to(DEV) if isinstance(v, torch.Tensor) else v) for k, v in inp.items()}
    with torch.no_grad():
        assert model.packed_epilogue
        a = model([inp])[0]
        model.packed_epilogue = False
        b = model([inp])[0]
    for key in ("output", "output_refined"):
        for x, y in zip(a[key], b[key]):
            assert (x - y).abs().max().item() < 2e-4, key   # log-DPV: the pooled image's last place through the network
    da = ops.dpv_expect(a["output_refined"][-1], inp["d_candi"], BV_log=True)
    db = ops.dpv_expect(b["output_refined"][-1], inp["d_candi"], BV_log=True)
    assert (da - db).abs().max().item() < 1e-3


# ---------------------------------------------------------------------------------------------------------------
# the uncertainty-field metric glue: compute_unc_field / compute_unc_rmse (utils/img_utils.py:178-202), fixture g19
# ---------------------------------------------------------------------------------------------------------------
def test_oracle_unc_field_metric_matches_the_reference():
    """The restatement against the reference's outputs on both dataset branches (kitti: shifted, no quash; ilim: unshifted,
    quashed): same ATen ops in the same order on the same host family -> equal fields, NaN columns included."""
    g = golden("g19_unc_field.npz")
    intr = torch.from_numpy(g["intr"])
    for tag in ("kitti", "ilim"):
        truth, pred, dbg = O.compute_unc_field(torch.from_numpy(g[tag + "_pred_logdpv"]), torch.from_numpy(g[tag + "_truth_dpv"]),
                                               g["d_candi"], intr, torch.from_numpy(g[tag + "_mask"]), tag)
        np.testing.assert_allclose(truth.numpy(), g[tag + "_field_truth"], rtol=0, atol=1e-7, equal_nan=True, err_msg=tag)
        np.testing.assert_allclose(pred.numpy(), g[tag + "_field_pred"], rtol=0, atol=1e-7, equal_nan=True, err_msg=tag)
        np.testing.assert_allclose(dbg.numpy(), g[tag + "_debugmap"], rtol=0, atol=1e-6, err_msg=tag)
        err = O.compute_unc_rmse(truth, pred, g["d_candi"])
        assert abs(float(err) - float(g[tag + "_rmse"])) < 1e-6, tag


@pytest.mark.gpu
def test_hip_unc_field_metric():
    """compute_unc_field / compute_unc_rmse of the package (two pdepth_ufield_f32 collapses + two pdepth_dpv_expect_f32) on the
    reference's inputs.  A pixel whose height or depth lies within ~1e-4 of a mask threshold may flip (this path sums E[d]
    in another order): the fields are compared on the columns the oracle finds stable under such a perturbation, the scalar
    error with a bound that a handful of flipped pixels cannot exceed."""
    from pdepth_amd.utils import img_utils
    g = golden("g19_unc_field.npz")
    intr = torch.from_numpy(g["intr"])
    for tag in ("kitti", "ilim"):
        cfg = synth.Cfg({"data": {"dataset_path": str(g[tag + "_path"])}})
        pred_v, truth_v, mask = (torch.from_numpy(g[tag + k]) for k in ("_pred_logdpv", "_truth_dpv", "_mask"))
        truth, pred, dbg = img_utils.compute_unc_field(pred_v.to(DEV), truth_v.to(DEV), g["d_candi"], intr.to(DEV), mask.to(DEV), cfg)
        for got, vol, want, kw in ((truth, truth_v, g[tag + "_field_truth"], dict(BV_log=False, mask=mask)),
                                   (pred, pred_v, g[tag + "_field_pred"], dict(BV_log=True))):
            stable = np.ones(want.shape[2], dtype=bool)
            okw = dict(O.UFIELD_DATASETS[tag], **kw)
            base, _ = O.gen_ufield(vol, g["d_candi"], intr[0], **okw)
            for eps in (-2e-5, 2e-5):
                p2, _ = O.gen_ufield(vol, g["d_candi"] * (1.0 + eps), intr[0], **okw)
                stable &= np.isclose(p2.numpy(), base.numpy(), rtol=1e-6, atol=1e-9, equal_nan=True).all(axis=(0, 1))
            assert stable.mean() > 0.8, tag
            np.testing.assert_allclose(got.cpu().numpy()[:, :, stable], want[:, :, stable], rtol=2e-5, atol=1e-6, equal_nan=True, err_msg=tag)
        err = float(img_utils.compute_unc_rmse(truth, pred, g["d_candi"]))
        assert abs(err - float(g[tag + "_rmse"])) < 5e-3 * max(1.0, float(g[tag + "_rmse"])), (tag, err, float(g[tag + "_rmse"]))
    with pytest.raises(UnboundLocalError):
        img_utils.gen_ufield(pred_v.to(DEV), g["d_candi"], intr[0].to(DEV), cfg=synth.Cfg({"data": {"dataset_path": "/data/other"}}))
