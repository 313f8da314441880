"""Phase stamps of the matrix-pipe sweep kernel: needs a library built with -DMFMA_STAMPS (PDEPTH_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
import pdepth_amd
from pdepth_amd import ops, synth, _native
pose = sys.argv[1] if len(sys.argv) > 1 else "mono"
B, C, D, H, W, V = 4, 67, 64, 256, 512, 1
if len(sys.argv) > 2 and sys.argv[2] == "cfg5":   # BASELINE config 5's shape (one volume)
    B, D, H, W, V = 1, 128, 512, 1024, 4
b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
for _ in range(2):
    ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="mfma")
torch.cuda.synchronize()

# the workspace tensor of the last call
ws = _native._last_workspace
tiles = ((W + 15) // 16) * ((H + 3) // 4)
flag_only = (B * tiles * 4 + 255) & ~255
q = ws[flag_only:flag_only + 256].cpu().view(torch.int64)   # 32 int64 = 64 ints; stamps start at int 16 = int64 index 8
st = q[8:20].tolist()
M = (1 << 64) - 1
ex = [int(x) & M for x in q[20:24].tolist()] + [int(q[19]) & M]
names = ["row setup", "positions", "row table", "X (MFMA)", "combine", "epilogue+rest", "passes", "failed trials", "blocks", "pixel rows"]
tot = sum(st[:6]) + st[10]
rows = st[9]
# stamps are s_memrealtime ticks (100 MHz): 10 ns each
if tot == 0: tot, rows = 1, 1   # (-DMFMA_EXIT_STAMPS alone: no phase stamps)
for n, v in zip(names[:6] + ["barrier+rotate"], st[:6] + [st[10]]): print("%-16s %6.1f %%   %7.2f us per sub-block" % (n, 100.0 * v / tot, v * 0.01 / rows))
print("sub-blocks %d  passes %.2f  failed trials %.2f  blocks %.2f  us per sub-block %.2f" % (rows, st[6] / rows, st[7] / rows, st[8] / rows, tot * 0.01 / rows))
first_exit, last_exit, n_wg, start = (~ex[0]) & M, ex[1], ex[3], (~ex[4]) & M
print("workgroups %d  kernel %.1f us  first exit %.1f  mean exit %.1f  -> idle tail %.1f %% of workgroup time" % (
    n_wg, (last_exit - start) * 0.01, (first_exit - start) * 0.01, (ex[2] / n_wg - start) * 0.01, 100.0 * (last_exit - ex[2] / n_wg) / (last_exit - start)))
