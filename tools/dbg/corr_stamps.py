"""Phase stamps of the correlation-form sweep kernel: needs a library built with -DCORR_STAMPS (tools/variants_corr.sh, PDEPTH_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
pose = sys.argv[1] if len(sys.argv) > 1 else "mono"
B, C, D, H, W, V = 4, 67, 64, 256, 512, 1
if len(sys.argv) > 2 and sys.argv[2] == "cfg5":   # BASELINE config 5's shape (one volume)
    B, D, H, W, V = 1, 128, 512, 1024, 4
if len(sys.argv) > 2 and sys.argv[2] == "small":
    B, H, W = 1, 64, 128
b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
for _ in range(3):
    ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="corr")
torch.cuda.synchronize()
ws = _native._last_workspace
tiles = ((W + 15) // 16) * ((H + 3) // 4)
flag_only = (B * tiles * 4 + 255) & ~255
q = ws[flag_only:flag_only + 256].cpu().view(torch.int64)   # stamps start at int 16 = int64 index 8
st = q[8:20].tolist()
names = ["queue: publish + barrier", "item set-up, pixel loads issued", "(ray wait) positions", "table atomics, (ref wait) centring", "barrier: tables",
         "scan, slots, Rr from LDS", "X: loads + MFMA", "wait Gram DMA", "barrier: X complete", "combine", "epilogue: stores, partial softmax",
         "barrier + merge + stores"]
tot = sum(st)
nblk = B * tiles * 4 * V * ((D + 63) // 64)   # passes
waves = 4
print("%s  B=%d %dx%d D=%d V=%d: wave time per pass %.2f us" % (pose, B, H, W, D, V, tot * 0.01 / (nblk * waves)))
for n, v in zip(names, st):
    print("  %-36s %5.1f %%  %6.2f us per pass and wave" % (n, 100.0 * v / tot, v * 0.01 / (nblk * waves)))
