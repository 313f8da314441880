"""Phase cycle breakdown of the cell-list kernel (library built with -DCELLS_STAMPS; PDEPTH_LIB selects it)."""
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import pdepth_amd
from pdepth_amd import ops, synth, _native
dev = torch.device("cuda:0")
pose = sys.argv[1] if len(sys.argv) > 1 else "mono"
H, W, D, V, B = 256, 512, 64, 1, 4
if len(sys.argv) > 2 and sys.argv[2] == "c5": H, W, D, V, B = 512, 1024, 128, 4, 2
b = synth.make_batch(2, B, C=67, D=D, H=H, W=W, V=V, pose=pose)
d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
for _ in range(3):
    out = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0)
torch.cuda.synchronize()
ws = _native._last_workspace
tiles = ((W + 15) // 16) * ((H + 3) // 4)
flag_only = (B * tiles * 4 + 255) & ~255
st = ws[flag_only + 64: flag_only + 64 + 104].view(torch.int64).cpu().numpy().astype(np.float64)
names = ["tile start", "positions", "scan+exchange", "cells/addr/stage0", "chunk: X compute", "barrier+dump", "plane loop", "epilogue", "tile-end barrier", "loop top", "chunk: DMA wait", "chunk: barrier", "chunk: DMA issue"]
tot = st.sum()
print(f"pose {pose} {H}x{W} D={D} V={V} B={B}: total wave-cycles {tot:.3e}")
for n, v in zip(names, st):
    print(f"  {n:22s} {v:.3e}  {100 * v / tot:5.1f} %")
