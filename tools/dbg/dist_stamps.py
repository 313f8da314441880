"""Phase stamps of the distance-form sweep kernel (library built with -DDIST_STAMPS): cycles per phase, per wave and pass."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
NAMES = ["queue", "set-up+loads issued", "positions", "table+centring", "B1", "operands+scan", "slots+X", "wait Q", "B2", "combine", "stores+softmax", "B3+merge+stores"]
for pose in sys.argv[1:] or ["mono", "stereo"]:
    B, C, D, H, W, V = (int(x) for x in os.environ.get("STAMP_SHAPE", "4,67,64,256,512,1").split(","))
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    for _ in range(3):
        ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo="dist")
    torch.cuda.synchronize()
    ws = _native._last_workspace
    n = B * ((W + 15) // 16) * ((H + 3) // 4)
    flag_only = (4 * n + 255) & ~255
    acc = ws[flag_only + 32: flag_only + 32 + 96].view(torch.int64).cpu().tolist()
    passes = B * H * W // 16 * V
    tot = sum(acc)
    print("%s: cycles per wave and pass (4 waves x %d passes), total %.0f" % (pose, passes, tot / (4 * passes)))
    for nm, a in zip(NAMES, acc):
        print("   %-22s %8.0f  %5.1f %%" % (nm, a / (4 * passes), 100.0 * a / tot))
