"""Runs the secondary kernels a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/dbg/next_rows_prof.py`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pdepth_amd
from pdepth_amd import ops, synth

B, D, H, W = 4, 64, 256, 512
x = torch.randn(B, D, H, W, device="cuda")
lp = torch.log_softmax(x, dim=1)
dc = ops.d_candi_tensor(synth.powerf(5, 40, D, 1.0), "cuda")
intr = torch.tensor([[0.58 * W, 0, W / 2.0, 0, 0.58 * W, H / 2.0, 0, 0, 1]], device="cuda").repeat(B, 1).reshape(B, 3, 3)
mk = (torch.rand(B, 1, H, W, device="cuda") > 0.6).float()
dm = (torch.rand(B, H, W, device="cuda") * 30 + 6) * mk[:, 0]
x1 = torch.randn(4, 64, 64, 128, device="cuda")
x2 = torch.randn(4, 64, 64, 128, device="cuda")
go = torch.randn(4, 81, 64, 128, device="cuda")
for _ in range(10):
    ops.ufield(lp, dc, intr, None, BV_log=True)
    ops.dpv_fuse(lp, dm, mk, dc, 0.3)
    ops.dpv_reduce_ex(x, dc, want_logp=True, want_depth=True, want_var=True, want_quarter=True)
    pdepth_amd._native.correlation_forward(x1, x2, 4, 1, 4, 1, 1, 1)
    pdepth_amd._native.correlation_backward(x1, x2, go, 4, 1, 4, 1, 1, 1)
torch.cuda.synchronize()
