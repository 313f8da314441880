"""Dump / compare the outputs of a few NCHW calls bit for bit across two builds of the library:
   PDEPTH_LIB=old.so python tools/dbg/same_bits.py dump ; python tools/dbg/same_bits.py check"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import pdepth_amd
from pdepth_amd import ops, synth
out = {}
for name, kw in (("b4", dict(B=4, C=67, D=64, H=256, W=512, V=1, pose="mono")), ("small", dict(B=1, C=67, D=64, H=64, W=128, V=1, pose="mono")),
                 ("odd", dict(B=2, C=22, D=83, H=37, W=53, V=3, pose="wide")), ("v4", dict(B=2, C=67, D=128, H=96, W=160, V=4, pose="mono")),
                 ("tiny", dict(B=3, C=5, D=7, H=4, W=16, V=2, pose="stereo"))):
    B = kw.pop("B")
    b = synth.make_batch(11, B, **kw)
    if name == "odd":
        b["ref"] += 3.0; b["src"] += 3.0
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    for algo in ("auto", "tiled1"):
        c, l, z = ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo, want_cost=True)
        out[name + "_" + algo] = (c.cpu(), l.cpu(), z.cpu())
path = os.path.join(REPO, "gpurun_out", "same_bits.pt")
if sys.argv[1] == "dump":
    torch.save(out, path)
    print("dumped", len(out))
else:
    ref = torch.load(path)
    bad = [k for k in out if not all(torch.equal(a.nan_to_num(), b.nan_to_num()) for a, b in zip(out[k], ref[k]))]
    print("same bits" if not bad else "DIFFERENT: %s" % bad)
