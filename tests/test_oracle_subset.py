"""The per-pixel form of the oracle's sweep (oracle.sweep_cost_at: what the full-size config-5 GPU test compares against)
against the whole-image form it restates, on sizes both finish in seconds."""
import numpy as np
import torch

import pdepth_amd  # noqa: F401
from pdepth_amd import synth
from oracle import ref_cpu as O


def test_sweep_cost_at_equals_sweep_cost():
    for seed, kw in ((1, dict(C=67, D=64, H=24, W=40, V=1, pose="mono")), (2, dict(C=19, D=33, H=17, W=29, V=3, pose="wide")),
                     (3, dict(C=67, D=16, H=16, W=48, V=2, pose="stereo"))):
        it = synth.make_item(seed, **kw)
        K = it["K"]
        cx, cy = K.numpy()[0, 2], K.numpy()[1, 2]
        for metric in ("L2", "L1"):
            full = O.sweep_cost(it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], K, it["rays"], cx, cy, 10.0, metric)
            hw = kw["H"] * kw["W"]
            idx = torch.from_numpy(np.random.default_rng(seed).permutation(hw)[: hw // 3]).long()
            at = O.sweep_cost_at(it["ref"][None], it["src"][None], it["d_candi"], it["R"], it["t"], K, it["rays"], cx, cy, 10.0, idx, metric)
            want = full.reshape(1, -1, hw)[:, :, idx]
            # same positions, same taps; the channel sum runs over another memory layout: an ulp or two of the cost
            np.testing.assert_allclose(at.numpy(), want.numpy(), rtol=2e-6, atol=1e-5)
