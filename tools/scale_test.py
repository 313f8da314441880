import sys; sys.path.insert(0,'.')
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth
dev=torch.device('cuda')
for pose in ('mono','stereo'):
  for peaked in (False, True):
    for scale in (1.0, 3.0, 10.0, 30.0):
        b=synth.make_batch(5,1,C=67,D=64,H=64,W=128,V=1,pose=pose,peaked=peaked)
        d={k:(v.to(dev) if isinstance(v,torch.Tensor) else v) for k,v in b.items()}
        ref=d['ref']*scale; src=d['src']*scale
        args=(ref,src,d['K'],d['R'],d['t'],d['rays'],d['cxcy'],d['d_candi'],10.0)
        ca,la,da=ops.sweep_dpv(*args,want_cost=True,algo='auto')
        cd,ld,dd=ops.sweep_dpv(*args,want_cost=True,algo='direct')
        # fp64 truth via torch double on GPU? use direct as reference
        print(f"{pose} peaked={peaked} scale={scale}: cost max {float(cd.abs().max()):.3g} | auto-direct cost {float((ca-cd).abs().max()):.3e} rel {float((ca-cd).abs().max()/cd.abs().max()):.2e} depth {float((da-dd).abs().max()):.3e}")
