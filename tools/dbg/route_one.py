"""One routed call shape for a kernel trace: python tools/dbg/route_one.py B H W off steps"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import pdepth_amd
from pdepth_amd import ops, synth
B, H, W, off, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
b = synth.make_batch(2, B, C=67, D=64, H=H, W=W, V=1, pose="mono")
mu = (torch.rand(67, generator=torch.Generator().manual_seed(5)) * 2 - 1) * off
b["ref"] += mu[None, :, None, None]; b["src"] += mu[None, None, :, None, None]
d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
dc = ops.d_candi_tensor(d["d_candi"], "cuda")
for algo in ("auto", "direct"):
    for _ in range(steps):
        ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo)
    torch.cuda.synchronize()
