"""When do the persistent blocks of the tiled kernel run out of work?  Needs a -DPDEPTH_EXIT_STAMPS build (PDEPTH_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
pose = sys.argv[1] if len(sys.argv) > 1 else "mono"
algo = sys.argv[2] if len(sys.argv) > 2 else "tiled2"
B, C, D, H, W, V = 4, 67, 64, 256, 512, 1
b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
for _ in range(3):
    ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
torch.cuda.synchronize()
ws = _native._last_workspace
tiles = ((W + 15) // 16) * ((H + 3) // 4)
flag_only = (B * tiles * 4 + 255) & ~255
q = ws[flag_only:flag_only + 256].cpu().view(torch.int64)
M = (1 << 64) - 1
u = [int(x) & M for x in q[8:21].tolist()]
first_exit, last_exit, n = (~u[0]) & M, u[1], u[3]
start = (~u[4]) & M
mean_exit = u[2] / n   # (sum of ~2^45-sized stamps x 512 blocks: no overflow)
print("%s %s: blocks %d  kernel %.1f us  first exit %.1f us  mean exit %.1f us  last exit %.1f us  -> idle tail %.1f %% of block time" % (
    pose, algo, n, (last_exit - start) * 0.01, (first_exit - start) * 0.01, (mean_exit - start) * 0.01, (last_exit - start) * 0.01,
    100.0 * (last_exit - mean_exit) / (last_exit - start)))
print("last exit per XCD (us): " + " ".join("%.1f" % ((x - start) * 0.01) for x in u[5:13]))
# per-item durations (the stamps build writes them over the depth of the item's first pixel)
_, _, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
torch.cuda.synchronize()
iw = 32 if algo == "tiled2" else 16
dur = depth[:, ::4, ::iw].cpu() * 0.01          # us, [B, 64, W/iw]
t0 = depth[:, ::4, 1::iw].cpu()
print("item duration us: mean %.1f  min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (dur.mean(), dur.min(), dur.median(), dur.flatten().kthvalue(int(0.9 * dur.numel())).values, dur.max()))
print("mean duration per half-band (16 strips, top to bottom), all items and columns:")
print(" ".join("%.0f" % dur[:, 4 * i:4 * i + 4, :].mean() for i in range(16)))
print("mean duration per column:")
print(" ".join("%.0f" % dur[:, :, c].mean() for c in range(dur.shape[2])))
print("mean duration per batch item: " + " ".join("%.0f" % dur[i].mean() for i in range(dur.shape[0])))
