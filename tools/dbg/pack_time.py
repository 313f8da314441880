"""Times the pre-pass (pdepth_pack_source_f32) with the library PDEPTH_LIB selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_kernels import timeit
print(os.environ.get("PDEPTH_LIB", "product library"))
for (B, V, H, W) in ((4, 1, 256, 512), (2, 4, 512, 1024), (4, 1, 64, 128), (1, 2, 61, 83)):
    src = torch.randn(B, V, 67, H, W, device="cuda")
    ms = min(timeit(lambda: ops.pack_source(src, 64), steps=30) for _ in range(3))
    by = B * V * H * W * 4 * (67 + 19 * 4)
    print("pack_source B=%d V=%d %dx%d: %.4f ms  %.0f GB/s" % (B, V, H, W, ms, by / ms * 1e-6), flush=True)
