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
