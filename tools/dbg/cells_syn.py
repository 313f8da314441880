import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import pdepth_amd
from pdepth_amd import ops, synth
from oracle import ref_cpu as O
dev = torch.device("cuda:0")
def run(C, D, H, W, V, pose, B=1):
    b = synth.make_batch(7, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    cost, logp, depth = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], d["d_candi"], 10.0, want_cost=True)
    torch.cuda.synchronize()
    for i in range(B):
        K = b["K"][i]
        oc, ol, od = O.sweep_dpv(b["ref"][i:i+1], b["src"][i:i+1], b["d_candi"], b["R"][i], b["t"][i], K, b["rays"][i], K.numpy()[0,2], K.numpy()[1,2], 10.0)
        e = (cost[i:i+1].cpu() - oc).abs()[0]
        bad = e > 2e-4 + 2e-5 * oc[0].abs()
        rows = bad.any(0).any(1).nonzero().flatten().tolist()
        print(f"C={C} D={D} {H}x{W} V={V} {pose} item {i}: bad {int(bad.sum())}/{bad.numel()} max {float(e.max()):.3e} depth err {float((depth[i:i+1].cpu()-od).abs().max()):.3e} rows {rows[:40]}")
for cfg in [(4,8,16,32,1,"mono"),(8,8,16,32,1,"mono"),(12,8,16,32,1,"mono"),(7,8,16,32,2,"mono"),(7,8,16,24,1,"stereo"),(67,64,32,64,1,"mono"),(67,64,64,128,1,"stereo"),(16,64,64,128,2,"mono")]:
    run(*cfg)
