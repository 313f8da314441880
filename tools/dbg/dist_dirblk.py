import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, pdepth_amd
from pdepth_amd import ops, synth, _native
tot = 0
for rep in range(6):
    B, H, W = 4, 256, 512
    b = synth.make_batch(2 + rep, B, C=67, D=64, H=H, W=W, V=1, pose="stereo")
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="dist")
    torch.cuda.synchronize()
    tot += _native._queue_slot(B, H, W, 59)
print("direct blocks over 6 stereo launches:", tot)
