"""One nmode's 5-frame trajectory a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/dbg/model_prof.py default`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import harness, synth
from pdepth_amd.models import get_model
nmode = sys.argv[1] if len(sys.argv) > 1 else "default"
model = get_model(synth.default_cfg(nmode), 0)
synth.seed_weights(model, seed=8)
model = model.cuda().eval()
frames = [harness.move_input(synth.make_model_input(4000 + i, B=1, V=1, H=256, W=512, D=64, pose="mono"), "cuda") for i in range(5)]
harness.eval_trajectory(model, frames)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(4):
    harness.eval_trajectory(model, frames)
torch.cuda.synchronize()
print("ms per frame", (time.perf_counter() - t0) / 20 * 1e3)
