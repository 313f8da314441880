"""dpv_fuse in the Gaussian-underflow regime: sparse depth 7.5 m away from the nearest candidate (sigma^2 = 0.3)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops
from oracle import ref_cpu as O
dc = np.array([0.0, 100.0, 200.0])
logp = torch.log_softmax(torch.zeros(1, 3, 1, 4), dim=1)
dm = torch.tensor([[[7.0, 7.5, 7.8, 8.2]]])
mk = torch.ones(1, 1, 1, 4)
print("oracle tofuse", O.gen_dpv_withmask(dm, mk, dc, 0.3)[0, :, 0].numpy())
print("oracle fused ", O.dpv_fuse(logp, dm, mk, dc, 0.3)[0][0, :, 0].numpy())
f, l = ops.dpv_fuse(logp.cuda(), dm.cuda(), mk.cuda(), dc, var=0.3)
print("hip fused    ", f[0, :, 0].cpu().numpy())
print("torch cuda exp(-93.75)", torch.exp(torch.tensor(-93.75).cuda()).item(), "cpu", torch.exp(torch.tensor(-93.75)).item())
