import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import pdepth_amd
from pdepth_amd import ops, synth, _native
from oracle import ref_cpu as O
dev = torch.device("cuda:0")
C, D, H, W, V, pose = 67, 64, 32, 64, 1, "mono"
b = synth.make_batch(7, 1, C=C, D=D, H=H, W=W, V=V, pose=pose)
d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0, want_cost=True)
torch.cuda.synchronize()
print("flagged tiles:", _native.fallback_tiles(1, H, W))
K = b["K"][0]
oc, ol, od = O.sweep_dpv(b["ref"][0:1], b["src"][0:1], b["d_candi"], b["R"][0], b["t"][0], K, b["rays"][0], K.numpy()[0,2], K.numpy()[1,2], 10.0)
e = (cost[0:1].cpu() - oc).abs()[0]
bad = e > 2e-4 + 2e-5 * oc[0].abs()
print("bad per plane:", [int(bad[k].sum()) for k in range(D)])
print("bad per row  :", [int(bad[:, y].sum()) for y in range(H)])
print("bad per col  :", [int(bad[:, :, x].sum()) for x in range(W)])
k = int(bad.reshape(D, -1).sum(1).argmax())
print("plane", k); 
for y in range(H): print("".join("X" if v else "." for v in bad[k, y]))
print("sample: got", cost[0, k, 0, :6].cpu().numpy(), "want", oc[0, k, 0, :6].numpy())
