"""Soak test on the GPU box: random shapes / poses / depth candidates / metrics, every implementation forced in turn
(both builds of the tiled kernel, the cell-list kernels where they apply: L2, D <= 128), against the gather kernel.
    python tools/soak.py <seed> <cases>   (1500 cases: worst relative difference 5.0e-7)"""
import sys, os; sys.path.insert(0,'.')
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
dev=torch.device('cuda')
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
worst=0; n=0; fb=0
for case in range(int(sys.argv[2]) if len(sys.argv)>2 else 200):
    algo=('tiled1','tiled2','cells')[case%3]
    H,W=int(rng.integers(2,200)),int(rng.integers(2,400)); C,D,V=int(rng.integers(1,72)),int(rng.integers(1,161)),int(rng.integers(1,4))
    B=int(rng.integers(1,3))
    pose=('mono','stereo','wide','identity')[int(rng.integers(0,4))]
    b=synth.make_batch(5000+case,B,C=C,D=D,H=H,W=W,V=V,pose=pose,cx_off=float(rng.uniform(-3,3)),cy_off=float(rng.uniform(-2,2)))
    k=int(rng.integers(0,6))
    if k==1:
        ang=rng.uniform(-0.3,0.3); cz,sz=np.cos(ang),np.sin(ang)
        b['R'][0,0]=torch.tensor([[cz,-sz,0],[sz,cz,0],[0,0,1]],dtype=torch.float32)@b['R'][0,0]
        b['t'][0,0]=torch.from_numpy(rng.uniform(-2.5,2.5,size=3).astype(np.float32))
    elif k==2: b['t'][0,0]=torch.from_numpy(rng.uniform(-30,30,size=3).astype(np.float32))
    elif k==3: b['d_candi']=rng.uniform(0.5,60.0,size=D)
    elif k==4: b['d_candi']=np.sort(rng.uniform(0.5,60.0,size=D))[::-1].copy()
    d={kk:(v.to(dev) if isinstance(v,torch.Tensor) else v) for kk,v in b.items()}
    metric='L1' if case%7==3 else 'L2'
    if algo=='cells' and (metric=='L1' or D>128): algo='tiled1'
    if algo=='tiled2' and D>64: algo='tiled1'
    ca=ops.sweep_cost(d['ref'],d['src'],d['K'],d['R'],d['t'],d['rays'],d['cxcy'],d['d_candi'],8.0,feat_dist=metric,algo=algo).cpu().numpy(); fb+=_native.fallback_tiles(B,H,W)
    cd=ops.sweep_cost(d['ref'],d['src'],d['K'],d['R'],d['t'],d['rays'],d['cxcy'],d['d_candi'],8.0,feat_dist=metric,algo='direct').cpu().numpy()
    if not np.array_equal(np.isnan(ca),np.isnan(cd)): print('NaN pattern differs',case,pose,H,W,C,D,V,B,k,metric); continue
    fin=np.isfinite(cd)
    if fin.any():
        err=float(np.abs(ca-cd)[fin].max())/max(1.0,float(np.abs(cd[fin]).max())); worst=max(worst,err)
        if err>2e-6: print('case',case,'variant',algo,pose,H,W,C,D,V,B,k,metric,'err',err)
    n+=1
print('cases',n,'worst',worst,'fallback tiles',fb)
