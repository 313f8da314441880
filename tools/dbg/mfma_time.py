"""Times the fused sweep through algo='mfma' with the library PDEPTH_LIB selects (A/B of experiment builds)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
def tm(name, B, C, D, H, W, V, pose, algo="mfma", steps=20):
    b = synth.make_batch(2, B, C=C, D=D, H=H, W=W, V=V, pose=pose)
    d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
    dc = ops.d_candi_tensor(d["d_candi"], "cuda")
    ms = min(timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0, algo=algo), steps=steps) for _ in range(3))
    print("%-28s %-7s %8.4f ms" % (name, algo, ms), flush=True)
print(os.environ.get("PDEPTH_LIB", "product library"))
tm("cfg2 mono 256x512", 4, 67, 64, 256, 512, 1, "mono")
tm("cfg3 stereo 256x512", 4, 67, 64, 256, 512, 1, "stereo")
if "--all" in sys.argv:
    tm("model-real 64x128", 4, 67, 64, 64, 128, 1, "mono", steps=50)
    tm("cfg5 D=128 512x1024 V=4", 2, 67, 128, 512, 1024, 4, "mono", steps=5)
