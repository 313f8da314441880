"""Seeded synthetic inputs for the sweep hot path.

Host-side numpy only.  Used by tests/, bench.py and the harness; the same seed gives the
same inputs for the HIP path and for the CPU oracle.  seed = 1000*config + item index.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .warping import view as _view

HFOV, VFOV = 80.0, 35.0  # KITTI-like field of view used for every synthetic camera


def powerf(d_min, d_max, n_depth, power):
    x = np.power(np.linspace(start=0, stop=1, num=n_depth), power)
    return np.array([d_min + (d_max - d_min) * v for v in x])


def _rot_yx(yaw, pitch):
    cy, sy = math.cos(yaw), math.sin(yaw)
    cp, sp = math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return Ry @ Rx


def make_pose(kind, rng):
    """(R [3,3], t [3]) float64.  'mono': ~0.8 m forward + <=1 deg rotation; 'stereo': KITTI baseline."""
    if kind == "stereo":
        return np.eye(3), np.array([0.54, 0.0, 0.0])
    if kind == "mono":
        ang = np.deg2rad(rng.uniform(-1.0, 1.0, size=2))
        t = np.array([0.02, 0.0, 0.8]) + rng.uniform(-0.05, 0.05, size=3)
        return _rot_yx(ang[0], ang[1]), t
    if kind == "wide":  # wide baseline + forward motion: tens of pixels of disparity per depth plane
        ang = np.deg2rad(rng.uniform(-2.0, 2.0, size=2))
        return _rot_yx(ang[0], ang[1]), np.array([2.5, 0.3, 1.5]) + rng.uniform(-0.1, 0.1, size=3)
    if kind == "identity":
        return np.eye(3), np.zeros(3)
    raise ValueError(kind)


def make_item(seed, C=67, D=64, H=64, W=128, V=1, pose="mono", peaked=False, cx_off=0.0, cy_off=0.0):
    """One batch item as CPU tensors (fp32 unless noted).

    Returns dict: ref [C,H,W], src [V,C,H,W], K [3,3], R [V,3,3], t [V,3], rays [3,HW],
    cxcy [2], d_candi float64 [D].
    """
    rng = np.random.default_rng(seed)
    ref = rng.standard_normal((C, H, W), dtype=np.float32)
    src = rng.standard_normal((V, C, H, W), dtype=np.float32)
    if peaked:
        for v in range(V):
            src[v] = 0.7 * np.roll(ref, shift=-(3 + v), axis=2) + 0.3 * src[v]
    cam = _view.camera_from_fov(W, H, HFOV, VFOV)
    K = cam["intrinsic_M"].copy()
    K[0, 2] += cx_off
    K[1, 2] += cy_off
    Rs, ts = [], []
    for v in range(V):
        R, t = make_pose(pose, rng)
        if pose == "stereo" and v > 0:
            t = t * (v + 1)
        Rs.append(R)
        ts.append(t)
    K32 = K.astype(np.float32)
    return {
        "ref": torch.from_numpy(ref), "src": torch.from_numpy(src),
        "K": torch.from_numpy(K32),
        "R": torch.from_numpy(np.stack(Rs).astype(np.float32)),
        "t": torch.from_numpy(np.stack(ts).astype(np.float32)),
        "rays": cam["unit_ray_array_2D"],
        "cxcy": torch.from_numpy(np.array([K32[0, 2], K32[1, 2]], dtype=np.float32)),
        "d_candi": powerf(5.0, 40.0, D, 1.0),
    }


def make_batch(config_id, B, first_item=0, **kw):
    """Stack B items (seed = 1000*config_id + item) into batched CPU tensors."""
    items = [make_item(1000 * config_id + first_item + i, **kw) for i in range(B)]
    out = {k: torch.stack([it[k] for it in items]) for k in ("ref", "src", "K", "R", "t", "rays", "cxcy")}
    out["d_candi"] = items[0]["d_candi"]
    return out


def seed_weights(model, seed=0, gain=0.5):
    """Deterministic weights keyed by parameter NAME (no checkpoint is available offline).

    Every floating-point state-dict entry is regenerated from a generator seeded with
    (seed, crc32(name)), so two implementations with the same parameter names and shapes -- this
    package's host model and the reference's -- end up with identical weights regardless of the order
    their constructors consumed the global RNG.  Convolution weights ~ gain * N(0, 2/fan_out) (the
    reference's init rule, damped: with BatchNorm in eval mode on near-identity statistics the
    16-block residual trunk otherwise amplifies activations to ~1e5 and every DPV degenerates to
    one-hot), BatchNorm statistics close to identity.  Returns the number of tensors written.
    """
    import zlib

    def fill(name, t):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
        leaf = name.rsplit(".", 1)[-1]
        if t.dim() >= 3:  # conv / transposed conv weight
            n = t.shape[0]
            for k in t.shape[2:]:
                n *= k
            val = torch.randn(t.shape, generator=g) * (gain * math.sqrt(2.0 / n))
        elif leaf == "running_var":
            val = 1.0 + 0.1 * torch.rand(t.shape, generator=g)
        elif leaf == "weight":  # norm scale
            val = 1.0 + 0.05 * torch.randn(t.shape, generator=g)
        else:  # biases, running_mean
            val = 0.05 * torch.randn(t.shape, generator=g)
        t.copy_(val.to(t.dtype))

    n = 0
    with torch.no_grad():
        items = list(model.state_dict().items())
        # the reference keeps Base3D's residual blocks in a plain list (not in its state_dict)
        b3d = getattr(model, "based_3d", None)
        if b3d is not None and isinstance(getattr(b3d, "dres_modules", None), list):
            for i, m in enumerate(b3d.dres_modules):
                items += [("based_3d.dres_modules.%d.%s" % (i, k), v) for k, v in m.state_dict().items()]
        for name, t in items:
            if t.is_floating_point():
                fill(name, t)
                n += 1
    return n


class Cfg(dict):
    """Attribute-style dict (stand-in for the reference's EasyDict, which is not installed here)."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Cfg):
                self[k] = Cfg(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def default_cfg(nmode="default", ndepth=64, feature_dim=64, model_name="base"):
    """The hot-path keys of configs/default_mono.json."""
    return Cfg({"data": {"model_name": model_name},
                "var": {"sigma_soft_max": 10.0, "feature_dim": feature_dim, "nmode": nmode, "ndepth": ndepth,
                        "bn_avg": True, "d_min": 5.0, "d_max": 40.0, "qpower": 1.0}})


HOT_PATH_VAR_KEYS = ("ndepth", "d_min", "d_max", "qpower", "sigma_soft_max", "feature_dim", "nmode")


def cfg_from_json(path):
    """The reference's experiment file (train.py:34-37 reads `configs/*.json` into an EasyDict) -> Cfg.

    Same schema: sections `data` (model_name, ...) and `var` (ndepth, d_min, d_max, qpower, sigma_soft_max, feature_dim,
    nmode, stereo, img_size = [width, height], crop_w, t_win, bn_avg, ...); every other section is carried along untouched.
    The keys the hot path reads must be present (the reference would fail on first use: models/models.py:444-448,
    trainer/default_trainer.py:38-41); optional ones get the reference's defaults (`stereo` absent = monocular)."""
    import json
    with open(path) as f:
        raw = json.load(f)
    for sec in ("data", "var"):
        if not isinstance(raw.get(sec), dict):
            raise KeyError("config %s: missing section '%s'" % (path, sec))
    missing = [k for k in HOT_PATH_VAR_KEYS if k not in raw["var"]]
    if missing:
        raise KeyError("config %s: var is missing %s" % (path, ", ".join(missing)))
    if "model_name" not in raw["data"]:
        raise KeyError("config %s: data.model_name is missing" % path)
    cfg = Cfg(raw)
    cfg.var.setdefault("bn_avg", True)
    cfg.var.setdefault("stereo", False)
    cfg.var.setdefault("t_win", 1)
    return cfg


def sweep_workload(cfg, dw=4):
    """What a config means for the sweep: channel count (feature_dim + the 3 pooled RGB channels, models/models.py:518-520),
    planes, depth candidates, the sweep resolution (the cropped image / dw: default_trainer.py:59-61, models.py:518) and
    the synthetic pose family that stands in for its data source (stereo rig or consecutive frames)."""
    v = cfg.var
    out = {"C": int(v.feature_dim) + 3, "D": int(v.ndepth), "sigma": float(v.sigma_soft_max),
           "pose": "stereo" if v.get("stereo") else "mono",
           "d_candi": powerf(float(v.d_min), float(v.d_max), int(v.ndepth), float(v.qpower))}
    if v.get("img_size") is not None:
        width = int(v.crop_w) if v.get("crop_w") else int(v.img_size[0])
        out.update({"image_hw": (int(v.img_size[1]), width), "H": int(v.img_size[1]) // dw, "W": width // dw})
    return out


def make_model_input(seed, B=1, V=1, H=256, W=256, D=64, pose="mono", dw=4):
    """The reference's model_input dict (kittiloader/batch_scheduler.py:147-283) with synthetic content.

    rgb [B,V+1,3,H,W] (reference view last), intrinsics [B,3,3] and unit_ray [B,3,h*w] at the 1/dw sweep
    resolution, src_cam_poses [B,V+1,4,4] (last = identity), d_candi float64, prev_output None.
    """
    rng = np.random.default_rng(seed)
    h, w = H // dw, W // dw
    cam = _view.camera_from_fov(w, h, HFOV, VFOV)
    K32 = cam["intrinsic_M"].astype(np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (B, V + 1, 1, 1))
    for b in range(B):
        for v in range(V):
            R, t = make_pose(pose, rng)
            poses[b, v, :3, :3] = R
            poses[b, v, :3, 3] = t
    base = rng.standard_normal((B, 1, 3, H, W), dtype=np.float32)
    rgb = 0.8 * base + 0.2 * rng.standard_normal((B, V + 1, 3, H, W), dtype=np.float32)
    return {
        "rgb": torch.from_numpy(rgb),
        "intrinsics": torch.from_numpy(np.tile(K32, (B, 1, 1))),
        "unit_ray": cam["unit_ray_array_2D"][None].repeat(B, 1, 1),
        "src_cam_poses": torch.from_numpy(poses),
        "d_candi": powerf(5.0, 40.0, D, 1.0),
        "prev_output": None,
    }
