"""Times pdepth_pack_source_f32 at the benchmark shape with the library PDEPTH_LIB selects."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth
from bench_kernels import timeit
b = synth.make_batch(2, 4, C=67, D=64, H=256, W=512, V=1, pose="mono")
src = b["src"].cuda()
ms = min(timeit(lambda: ops.pack_source(src, 64), steps=50) for _ in range(3))
print(os.environ.get("PDEPTH_LIB", "product"), "pack_source %.2f us" % (ms * 1e3))
