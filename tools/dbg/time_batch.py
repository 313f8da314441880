"""Per-volume time of the fused sweep against the batch size (how much of a launch is its tail / fixed cost?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
for pose in ("mono", "stereo"):
    for B in (1, 2, 4, 8, 16):
        b = synth.make_batch(2, B, C=67, D=64, H=256, W=512, V=1, pose=pose)
        d = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
        dc = ops.d_candi_tensor(d["d_candi"], "cuda")
        ps = ops.pack_source(d["src"], 64)
        ms = min(timeit(lambda: ops.sweep_dpv(d["ref"], d["src"], d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0), steps=20) for _ in range(3))
        mp = min(timeit(lambda: ops.sweep_dpv(d["ref"], ps, d["K"], d["R"], d["t"], d["rays"], d["cxcy"], dc, 10.0), steps=20) for _ in range(3))
        print(f"{pose:7s} B={B:2d}: {ms:7.4f} ms = {1e3 * ms / B:6.1f} us/volume; packed entry {mp:7.4f} ms = {1e3 * mp / B:6.1f} us/volume", flush=True)
