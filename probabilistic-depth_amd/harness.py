"""Evaluation harness: the counterpart of DefaultTrainer._validate_with_gt for the hot path.

Reproduces the model call + depth regression of the reference's eval loop
(trainer/default_trainer.py:171-321): ``model([inp])[0]`` -> ``prev_output = interpolate(output_refined[-1],
0.25, 'nearest')`` (:221) -> per item ``dpv_to_depthmap(output[-1][b])`` and
``dpv_to_depthmap(output_refined[-1][b])`` (:229-233) -- with the per-item Python loop replaced by one
batched expectation launch per resolution.  Dataset IO, ground-truth metrics and visualisation are out
of scope.
"""
import torch
import torch.nn.functional as F

from . import ops


def move_input(model_input, device):
    return {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in model_input.items()}


@torch.no_grad()
def eval_step(model, model_input, prev_output=None):
    """One frame of the eval loop.  Returns dict(output, depth_lowres [B,h,w], depth_refined [B,H,W],
    prev_output [B,D,h,w] for the next frame)."""
    model_input = dict(model_input)
    model_input["prev_output"] = prev_output
    out = model([model_input])[0]
    d_candi = model_input["d_candi"]
    # the host model's DPV passes leave the depth maps and the next prev_output behind (one pass over each volume
    # instead of three); any other model gets the reference's op sequence
    aux = getattr(model, "last_aux", None) or {}
    nxt = aux.get("prev_output")
    if nxt is None:
        nxt = F.interpolate(out["output_refined"][-1].detach(), scale_factor=0.25, mode="nearest")
    low, ref = aux.get("depth_lowres"), aux.get("depth_refined")
    return {
        "output": out,
        "depth_lowres": low if low is not None else ops.dpv_expect(out["output"][-1], d_candi, BV_log=True),
        "depth_refined": ref if ref is not None else ops.dpv_expect(out["output_refined"][-1], d_candi, BV_log=True),
        "prev_output": nxt,
    }


@torch.no_grad()
def eval_trajectory(model, frames):
    """Chained frames of one trajectory (prev_output fed back, reset at frame 0: default_trainer.py:200-202)."""
    prev, results = None, []
    for inp in frames:
        r = eval_step(model, inp, prev)
        prev = r["prev_output"]
        results.append(r)
    return results



def model_from_config(path, device, id=0):
    """get_model() for an experiment file in the reference's JSON schema (train.py:34-37 + models/get_model.py:4-13):
    returns (model on `device` in eval mode, cfg, the float64 depth candidates of default_trainer.py:38-39)."""
    from . import synth
    from .models import get_model
    cfg = synth.cfg_from_json(path)
    model = get_model(cfg, id).to(device).eval()
    return model, cfg, synth.sweep_workload(cfg)["d_candi"]
