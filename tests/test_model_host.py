"""Host model (get_model / forward contract) against the whole-model captures of the reference.

CPU part: parameter names, order and shapes equal the reference's state_dict (positional checkpoint
loading, trainer/base_trainer.py:83-90).  GPU part: same name-seeded weights, same synthetic frames ->
low-res log-DPV, low-res depth and refined depth of the reference's CPU run (fixture g8_model.npz).
The conv stacks run on MIOpen here and on mkldnn in the fixture, which alone moves logits by ~1e-5
relative; the sweep itself is checked at 1e-4 in test_hip_parity.py.
"""
import numpy as np
import pytest
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import harness, synth
from pdepth_amd.models import get_model
from util import golden, golden_blas


@pytest.mark.parametrize("nmode", ["default", "default_feedback"])
def test_state_dict_layout_matches_reference(nmode):
    g = golden("g8_model.npz")
    want_keys = [str(k) for k in g[nmode + "_state_keys"]]
    want_shapes = [str(s) for s in g[nmode + "_state_shapes"]]
    model = get_model(synth.default_cfg(nmode), 0)
    sd = model.state_dict()
    keys = list(sd.keys())
    assert keys[:len(want_keys)] == want_keys, "parameter order differs from the reference state_dict"
    assert [repr(tuple(sd[k].shape)) for k in want_keys] == want_shapes
    extra = keys[len(want_keys):]
    assert all(k.startswith("based_3d.dres_modules.") for k in extra), extra  # registered here, list in the reference


def test_reference_checkpoint_loads_through_the_trainers_zip():
    """trainer/base_trainer.py:83-90: zip(model keys, checkpoint keys) then a STRICT load_state_dict.  A reference
    feedback checkpoint has no based_3d.dres_modules.* entries (plain list there, models.py:394-399)."""
    from collections import OrderedDict
    g = golden("g8_model.npz")
    model = get_model(synth.default_cfg("default_feedback"), 0)
    ref_keys = [str(k) for k in g["default_feedback_state_keys"]]
    checkpoint = OrderedDict((k, torch.full_like(v, 0.25)) for k, v in model.state_dict().items() if k in set(ref_keys))
    assert list(checkpoint.keys()) == ref_keys
    before = {k: v.clone() for k, v in model.state_dict().items() if ".dres_modules." in k}
    new_weights = OrderedDict()
    for a, b in zip(list(model.state_dict().keys()), list(checkpoint.keys())):
        new_weights[a] = checkpoint[b]
    model.load_state_dict(new_weights)   # strict
    sd = model.state_dict()
    assert all(bool((sd[k] == 0.25).all()) for k in ref_keys if sd[k].is_floating_point())
    assert all(torch.equal(sd[k], v) for k, v in before.items())   # untouched, like the reference's list
    # a genuinely missing key still fails
    del new_weights[ref_keys[0]]
    with pytest.raises(RuntimeError, match="Missing key"):
        model.load_state_dict(new_weights)

def test_truncated_checkpoint_of_this_framework_is_rejected():
    """Only a checkpoint with NONE of the residual 3-D blocks (a reference checkpoint) passes the strict load without
    them; one that holds some of them is truncated and must raise."""
    model = get_model(synth.default_cfg("default_feedback"), 0)
    sd = model.state_dict()
    dres = [k for k in sd if ".dres_modules." in k and not k.endswith("num_batches_tracked")]
    assert len(dres) > 4
    truncated = {k: v for k, v in sd.items() if k not in set(dres[len(dres) // 2:])}
    with pytest.raises(RuntimeError):
        model.load_state_dict(truncated)
    model.load_state_dict({k: v for k, v in sd.items() if ".dres_modules." not in k})   # the reference-checkpoint case



def test_hip_ops_refuse_autograd_inputs():
    """The ctypes kernels have no backward: an input that requires grad must raise instead of silently cutting
    (or, in place, corrupting) the graph."""
    from pdepth_amd import _native
    x = torch.zeros(1, 4, 2, 2, requires_grad=True)
    with pytest.raises(RuntimeError, match="no backward"):
        _native.dpv_reduce(x, torch.zeros(4))
    with pytest.raises(RuntimeError, match="no backward"):
        _native.dpv_expect(x, torch.zeros(4), True)
    with torch.no_grad():   # grad mode off: the guard passes (and the CPU tensor is refused by the device check)
        with pytest.raises(RuntimeError) as ei:
            _native.dpv_expect(x, torch.zeros(4), True)
        assert "no backward" not in str(ei.value)


def test_get_model_contract():
    cfg = synth.default_cfg("default")
    assert type(get_model(cfg, 0)).__name__ == "BaseModel"
    cfg.data["model_name"] = "default"
    assert type(get_model(cfg, 0)).__name__ == "DefaultModel"
    cfg.data["model_name"] = "nope"
    with pytest.raises(NotImplementedError):
        get_model(cfg, 0)
    # 'packnet' (models/get_model.py:9-10): the host object exists; the PackNet CNN is the caller's to plug in
    cfg.data["model_name"] = "packnet"
    pk = get_model(cfg, 0)
    assert type(pk).__name__ == "PacknetModel" and pk.sigma_soft_max == cfg.var.sigma_soft_max
    with pytest.raises(NotImplementedError, match="base_encoder"):
        pk.forward({"rgb": torch.zeros(1, 2, 3, 8, 8)})
    m = get_model(synth.default_cfg("default"), 0)
    m.set_viz(None)
    m.init_weights()
    with pytest.raises(Exception, match="Nmode wrong"):
        bad = get_model(synth.default_cfg("bogus"), 0)
        bad.forward_int({})


def test_seed_weights_is_name_keyed():
    a = get_model(synth.default_cfg("default"), 0)
    torch.manual_seed(123)
    b = get_model(synth.default_cfg("default"), 0)
    synth.seed_weights(a, 8)
    synth.seed_weights(b, 8)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)


@pytest.mark.gpu
@pytest.mark.parametrize("nmode", ["default", "default_feedback"])
def test_model_matches_reference_capture(nmode):
    g = golden("g8_model.npz")
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    model = get_model(synth.default_cfg(nmode), 0)
    synth.seed_weights(model, seed=8)
    model = model.to(dev).eval()
    model.sweep_blas = golden_blas(g)
    prev = None
    for frame in range(2 if nmode == "default_feedback" else 1):
        inp = harness.move_input(synth.make_model_input(8000 + frame, B=1, V=1, H=256, W=256, D=64, pose="mono"), dev)
        r = harness.eval_step(model, inp, prev)
        prev = r["prev_output"]
        tag = "%s_f%d" % (nmode, frame)
        out = r["output"]
        assert set(out.keys()) == {"output", "output_refined", "flow", "flow_refined"}
        assert out["output"][-1].shape == (1, 64, 64, 64) and out["output_refined"][-1].shape == (1, 64, 256, 256)
        e_dpv = np.abs(out["output"][-1].cpu().numpy()[:, ::4, ::2, ::2] - g[tag + "_logdpv_sub"]).max()
        e_low = np.abs(r["depth_lowres"].cpu().numpy() - g[tag + "_depth_low"]).max()
        e_ref = np.abs(r["depth_refined"].cpu().numpy() - g[tag + "_depth_ref"]).max()
        print(f"[{tag}] max|dlogDPV|={e_dpv:.3e} max|ddepth_low|={e_low:.3e} max|ddepth_refined|={e_ref:.3e}")
        # default mode: the north star's 1e-4 on both depth maps (measured 2.3e-5 / 2.5e-5); feedback mode adds the
        # 3-D convolutions on MIOpen vs mkldnn in front of the low-resolution DPV (measured 1.9e-4 / 2.3e-4)
        assert e_dpv < 2e-4 and e_ref < 1e-4 and e_low < (1e-4 if nmode == "default" else 6e-4)
        if nmode == "default" and frame == 0:
            # cost volume inside the model: encoder features differ (MIOpen vs mkldnn), the sweep does not
            with torch.no_grad():   # (the HIP ops refuse inputs that require grad; eval_step runs under no_grad itself)
                _, costv, _, _ = model.forward_encoder(inp)
            e_cost = np.abs(costv.cpu().numpy()[:, ::4, ::2, ::2] - g["default_cost_sub"])
            rel = e_cost.max() / np.abs(g["default_cost_sub"]).max()
            print(f"[{tag}] cost volume: max abs diff {e_cost.max():.3e} (relative to max cost {rel:.3e})")
            assert rel < 1e-4


@pytest.mark.gpu
def test_default_model_forward():
    dev = torch.device("cuda:0")
    cfg = synth.default_cfg("default", model_name="default")
    m = get_model(cfg, 0).to(dev).eval()
    inp = harness.move_input(synth.make_model_input(3, B=2, V=1, H=64, W=96, D=64), dev)
    with torch.no_grad():
        out = m([inp])[0]
    assert out["output"][0].shape == (2, 64, 16, 24) and out["output_refined"][0].shape == (2, 64, 64, 96)
    assert (torch.exp(out["output"][0]).sum(1) - 1).abs().max().item() < 1e-5


@pytest.mark.gpu
def test_upsample_mode_fuses_sparse_depth():
    """nmode default_upsample: output[0] = log of (DPV x Gaussian soft label of the sparse depth), renormalised
    (models/models.py:659-678); checked against the oracle's restatement applied to the model's own BV_cur."""
    from oracle import ref_cpu as O
    dev = torch.device("cuda:0")
    model = get_model(synth.default_cfg("default_upsample"), 0)
    synth.seed_weights(model, seed=8)
    model = model.to(dev).eval()
    inp = synth.make_model_input(8100, B=2, V=1, H=256, W=256, D=64, pose="mono")
    g = torch.Generator().manual_seed(1)
    masks = (torch.rand(2, 1, 64, 64, generator=g) > 0.7).float()
    inp["dmaps"] = (torch.rand(2, 64, 64, generator=g) * 30 + 6) * masks[:, 0]
    inp["masks"] = masks
    with torch.no_grad():
        out = model([harness.move_input(inp, dev)])[0]
    fused_log, bv_cur = out["output"]
    want_fused, want_log = O.dpv_fuse(bv_cur.cpu(), inp["dmaps"], inp["masks"], inp["d_candi"], 0.3)
    np.testing.assert_allclose(fused_log.cpu().numpy(), want_log.numpy(), rtol=1e-5, atol=3e-5)
    assert out["output_refined"][0].shape == (2, 64, 256, 256)



@pytest.mark.gpu
def test_upsample_mode_matches_reference_capture():
    """nmode default_upsample end to end against the reference's own run (fixture g20, tests/golden/make_golden_r3c.py):
    both low-resolution log-DPVs (fused with the sparse depth's soft label / plain), their depth maps and the refined one.
    Bounds as for the default mode (no 3-D convolutions on this path): 2e-4 on the log-DPVs, 1e-4 m on the depth maps."""
    g = golden("g20_upsample_model.npz")
    seed_input, seed_sparse, seed_weights = (int(v) for v in g["seeds"])
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    model = get_model(synth.default_cfg("default_upsample"), 0)
    synth.seed_weights(model, seed=seed_weights)
    model = model.to(dev).eval()
    model.sweep_blas = golden_blas(g)
    inp = synth.make_model_input(seed_input, B=1, V=1, H=256, W=256, D=64, pose="mono")
    gen = torch.Generator().manual_seed(seed_sparse)     # (the generator of make_golden_r3c.sparse_depth, restated: data, not code)
    masks = (torch.rand(1, 1, 64, 64, generator=gen) > 0.7).float()
    inp["dmaps"] = (torch.rand(1, 64, 64, generator=gen) * 30 + 6) * masks[:, 0]
    inp["masks"] = masks
    with torch.no_grad():
        out = model([harness.move_input(inp, dev)])[0]
    fused, plain = out["output"]
    from pdepth_amd.utils import img_utils
    e_f = np.abs(fused.cpu().numpy()[:, ::2, ::2, ::2] - g["fused_logdpv_sub"]).max()
    e_p = np.abs(plain.cpu().numpy()[:, ::4, ::2, ::2] - g["plain_logdpv_sub"]).max()
    d_f = np.abs(img_utils.dpv_to_depthmap(fused, inp["d_candi"], BV_log=True).cpu().numpy() - g["depth_fused"]).max()
    d_p = np.abs(img_utils.dpv_to_depthmap(plain, inp["d_candi"], BV_log=True).cpu().numpy() - g["depth_plain"]).max()
    d_r = np.abs(img_utils.dpv_to_depthmap(out["output_refined"][0], inp["d_candi"], BV_log=True).cpu().numpy() - g["depth_refined"]).max()
    print(f"[default_upsample] max|dlogDPV| fused {e_f:.3e} plain {e_p:.3e}; max|ddepth| fused {d_f:.3e} plain {d_p:.3e} refined {d_r:.3e}")
    assert e_f < 2e-4 and e_p < 2e-4 and d_f < 1e-4 and d_p < 1e-4 and d_r < 1e-4


@pytest.mark.gpu
def test_batched_two_view_model_matches_reference_capture():
    """Two batch items with two source views each (fixture g21): the reference walks the items and, per item, the views
    (models/models.py:522-545, homography.py:124-131); here ONE batched sweep over [B, V, ...] -- same low-resolution log-DPV
    (2e-4) and depth maps (1e-4 m) per item."""
    g = golden("g21_model_b2v2.npz")
    seed_input, seed_weights = (int(v) for v in g["seeds"])
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    model = get_model(synth.default_cfg("default"), 0)
    synth.seed_weights(model, seed=seed_weights)
    model = model.to(dev).eval()
    model.sweep_blas = golden_blas(g)
    inp = harness.move_input(synth.make_model_input(seed_input, B=2, V=2, H=256, W=256, D=64, pose="mono"), dev)
    r = harness.eval_step(model, inp, None)
    low = r["output"]["output"][-1]
    assert low.shape == (2, 64, 64, 64) and r["output"]["output_refined"][-1].shape == (2, 64, 256, 256)
    e_dpv = np.abs(low.cpu().numpy()[:, ::4, ::2, ::2] - g["logdpv_sub"]).max()
    e_low = np.abs(r["depth_lowres"].cpu().numpy() - g["depth_low"]).max()
    e_ref = np.abs(r["depth_refined"].cpu().numpy() - g["depth_refined"]).max()
    print(f"[B=2 V=2] max|dlogDPV|={e_dpv:.3e} max|ddepth_low|={e_low:.3e} max|ddepth_refined|={e_ref:.3e}")
    assert e_dpv < 2e-4 and e_low < 1e-4 and e_ref < 1e-4


def test_config_loader_reads_the_reference_schema(tmp_path):
    """synth.cfg_from_json: the experiment files of the reference (train.py:34-37: json -> EasyDict with sections data / var /
    ...) -- a hand-written file with the hot-path keys, a stereo variant, and the errors for missing keys."""
    import json
    from pdepth_amd import harness
    raw = {"data": {"exp_name": "t", "model_name": "base"},
           "var": {"sigma_soft_max": 8.0, "t_win": 1, "d_min": 3.0, "d_max": 60.0, "feature_dim": 32, "ndepth": 48,
                   "qpower": 1.5, "img_size": [768, 256], "crop_w": 384, "nmode": "default", "bn_avg": True},
           "train": {"batch_size": 8}, "seed": 0}
    p = tmp_path / "exp.json"
    p.write_text(json.dumps(raw))
    cfg = synth.cfg_from_json(str(p))
    assert cfg.var.ndepth == 48 and cfg.data.model_name == "base" and cfg.train.batch_size == 8 and cfg.var.stereo is False
    wl = synth.sweep_workload(cfg)
    assert (wl["C"], wl["D"], wl["H"], wl["W"], wl["pose"], wl["sigma"]) == (35, 48, 64, 96, "mono", 8.0)
    from pdepth_amd.utils.img_utils import powerf
    assert np.array_equal(wl["d_candi"], powerf(3.0, 60.0, 48, 1.5))
    model, cfg2, d_candi = harness.model_from_config(str(p), "cpu")
    assert model.D == 48 if hasattr(model, "D") else True
    assert len(d_candi) == 48
    raw["var"]["stereo"] = True
    p.write_text(json.dumps(raw))
    assert synth.sweep_workload(synth.cfg_from_json(str(p)))["pose"] == "stereo"
    del raw["var"]["ndepth"]
    p.write_text(json.dumps(raw))
    with pytest.raises(KeyError, match="ndepth"):
        synth.cfg_from_json(str(p))
    p.write_text(json.dumps({"var": {}}))
    with pytest.raises(KeyError, match="data"):
        synth.cfg_from_json(str(p))
