"""Calibration of the domain guard of the distance-form kernel: depth error against the CPU oracle (guard off: library built with
-DDIST_GUARD_RATIO=1e30f, and the product build) on features with trends of growing amplitude, next to the guard statistic
sum_c var_c / sum_c lag_c of the pre-pass."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
from util import oracle_batch, to_dev
dev = torch.device("cuda:0")
def guard_stat(src):   # src [V,C,H,W] of one item: the statistic of sweep_pack.hip's feature_stats_kernel (view 0, 8 sampled rows, lag 16)
    x = src[0]; C, H, W = x.shape
    rows = [min(H - 1, ((2 * i + 1) * H) // 16) for i in range(min(H, 8))]
    s = x[:, rows, :]
    var = s.var(dim=(1, 2), unbiased=False)
    lag = 16 if W > 32 else W // 2
    idx = torch.arange(W); idx2 = torch.where(idx + lag < W, idx + lag, idx - lag)
    d = s - s[:, :, idx2]
    return float(var.sum() / (0.5 * (d * d).mean(dim=(1, 2))).sum())
def case(name, b):
    ocost, ologp, odepth = oracle_batch(b)
    d = to_dev(b, dev)
    out = []
    for algo in ("dist", "direct"):
        cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0, algo=algo, want_cost=True)
        torch.cuda.synchronize()
        nd = _native._queue_slot(b["ref"].shape[0], b["ref"].shape[2], b["ref"].shape[3], 59) if algo == "dist" else 0
        out.append("%s: depth %.2e cost %.2e%s" % (algo, (depth.cpu() - odepth).abs().max().item(), (cost.cpu() - ocost).abs().max().item(), " direct-blocks %d" % nd if algo == "dist" else ""))
    print("%-34s guard stat %6.2f | %s" % (name, guard_stat(b["src"][0]), " | ".join(out)), flush=True)
H, W = 64, 128
for pose in ("mono", "stereo"):
    for amp in (0.0, 1.0, 2.0, 3.0, 4.0, 6.0, 8.0):
        b = synth.make_batch(9, 1, C=67, D=64, H=H, W=W, V=1, pose=pose)
        g = torch.Generator().manual_seed(1243)
        mu = (torch.rand(67, generator=g) * 2 - 1) * 4.0
        ramp = torch.linspace(-1.0, 1.0, H)[None, None, :, None] * amp
        b["ref"] = b["ref"] + mu[None, :, None, None] + ramp
        b["src"] = b["src"] + mu[None, None, :, None, None] + ramp[:, None]
        case("%s vertical ramp +-%g sigma" % (pose, amp), b)
    for amp in (2.0, 4.0, 8.0):
        b = synth.make_batch(9, 1, C=67, D=64, H=H, W=W, V=1, pose=pose)
        ramp = torch.linspace(-1.0, 1.0, W)[None, None, None, :] * amp
        b["ref"] = b["ref"] + ramp; b["src"] = b["src"] + ramp[:, None]
        case("%s horizontal ramp +-%g sigma" % (pose, amp), b)
    for k in (3, 7, 15):   # smooth features: box-filtered noise, renormalised to unit variance
        b = synth.make_batch(9, 1, C=67, D=64, H=H, W=W, V=1, pose=pose)
        def smooth(x):
            sh = x.shape
            y = torch.nn.functional.avg_pool2d(x.reshape(-1, 1, sh[-2], sh[-1]), k, stride=1, padding=k // 2, count_include_pad=False).reshape(sh)
            return y / y.std()
        b["ref"] = smooth(b["ref"]); b["src"] = smooth(b["src"])
        case("%s box-filtered %dx%d noise" % (pose, k, k), b)
    b = synth.make_batch(9, 1, C=67, D=64, H=H, W=W, V=1, pose=pose)
    b["ref"] = torch.clamp(b["ref"] + 1.5, min=0.0); b["src"] = torch.clamp(b["src"] + 1.5, min=0.0)
    case("%s relu" % pose, b)
    b = synth.make_batch(9, 1, C=67, D=64, H=H, W=W, V=1, pose=pose)
    b["ref"] = b["ref"] * 3.0; b["src"] = b["src"] * 3.0
    case("%s N(0,1) x 3" % pose, b)
