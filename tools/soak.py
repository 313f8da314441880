"""Soak test on the GPU box: random shapes / poses / depth candidates / metrics, every implementation forced in turn
(both builds of the tiled kernel, the cell-list kernels where they apply: L2, D <= 128), against the gather kernel.
    python tools/soak.py <seed> <cases>   (1500 cases: worst relative difference 5.0e-7)"""
import sys, os; sys.path.insert(0,'.')
if os.environ.get("PDEPTH_LAX"):   # bisecting with libraries of older commits (PDEPTH_LIB): tolerate missing symbols / an older ABI number
    import ctypes
    _CDLL = ctypes.CDLL
    class _Dummy:
        argtypes = None; restype = None
        def __call__(self, *a): return 2
    class _Lax:
        def __init__(self, *a, **k): object.__setattr__(self, "_l", _CDLL(*a, **k))
        def __getattr__(self, n):
            if n == "pdepth_abi_version": return _Dummy()
            try: return getattr(self._l, n)
            except AttributeError: return _Dummy()
    ctypes.CDLL = _Lax
import numpy as np, torch
import pdepth_amd
from pdepth_amd import ops, synth, _native
dev=torch.device('cuda')
rng=np.random.default_rng(int(sys.argv[1]) if len(sys.argv)>1 else 1)
worst=0; worst_d=0; n=0; fb=0
SPEC=bool(os.environ.get('SOAK_SPEC'))   # the evaluation configuration's channel / plane / view counts at random image sizes and poses
DPV=bool(os.environ.get('SOAK_DPV'))   # also compare the fused log-DPV / depth outputs and the packed-source entry
only=set(int(x) for x in sys.argv[3].split(',')) if len(sys.argv)>3 else None   # replay: only these case numbers (the RNG is advanced through the others)
force=sys.argv[4] if len(sys.argv)>4 else None   # replay: force this implementation
for case in range(int(sys.argv[2]) if len(sys.argv)>2 else 200):
    algo=('tiled1','tiled2','cells')[case%3]
    H,W=int(rng.integers(2,200)),int(rng.integers(2,400)); C,D,V=int(rng.integers(1,72)),int(rng.integers(1,161)),int(rng.integers(1,4))
    B=int(rng.integers(1,3))
    if SPEC: C,D,V=67,64,1; H,W=int(rng.integers(40,300)),int(rng.integers(100,560))   # the specialised instantiation (two-tile build from 96 k pixels)
    pose=('mono','stereo','wide','identity')[int(rng.integers(0,4))]
    cxo,cyo=float(rng.uniform(-3,3)),float(rng.uniform(-2,2))
    k=int(rng.integers(0,6))
    if only is not None and case not in only:   # replay: consume this case's random draws without building it
        if k==1: rng.uniform(-0.3,0.3); rng.uniform(-2.5,2.5,size=3)
        elif k==2: rng.uniform(-30,30,size=3)
        elif k in (3,4): rng.uniform(0.5,60.0,size=D)
        continue
    b=synth.make_batch(5000+case,B,C=C,D=D,H=H,W=W,V=V,pose=pose,cx_off=cxo,cy_off=cyo)
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
    if case%4==3 and metric=='L2' and not force: algo='auto'
    if force: algo=force
    if DPV:   # the fused outputs too: log-DPV and expected depth of the implementation against the gather kernel's, and the packed entry
        args=(d['ref'],d['src'],d['K'],d['R'],d['t'],d['rays'],d['cxcy'],d['d_candi'],8.0)
        ca_,la,da=ops.sweep_dpv(*args,feat_dist=metric,algo=algo,want_cost=True)
        cd_,ld,dd=ops.sweep_dpv(*args,feat_dist=metric,algo='direct',want_cost=True)
        fd=torch.isfinite(dd)&torch.isfinite(da)
        if bool((torch.isfinite(dd)!=torch.isfinite(da)).any()): print('case',case,algo,'depth finiteness differs',pose,H,W,C,D,V,B,k,metric)
        elif bool(fd.any()):
            dmax=float(np.max(np.abs(b['d_candi']))); de=float((da-dd)[fd].abs().max())
            worst_d=max(worst_d,de/max(1.0,dmax/40.0))
            if de>3e-4*max(1.0,dmax/40.0): print('case',case,'variant',algo,pose,H,W,C,D,V,B,k,metric,'depth differs by',de,'max candidate',dmax)
        if case%5==0 and algo in ('auto','tiled1','tiled2') and metric=='L2' and C<=68:
            try:
                ps=ops.pack_source(d['src'],D)
                cp,lp,dp=ops.sweep_dpv(d['ref'],ps,*args[2:],feat_dist=metric,algo='auto',want_cost=True)
                ca2,la2,da2=ops.sweep_dpv(*args,feat_dist=metric,algo='auto',want_cost=True)
                if not (torch.equal(cp.nan_to_num(),ca2.nan_to_num()) and torch.equal(dp.nan_to_num(),da2.nan_to_num())): print('case',case,'packed entry differs from the plain entry',pose,H,W,C,D,V,B,k)
            except RuntimeError as e:
                if 'packed' not in str(e): raise
    ca=ops.sweep_cost(d['ref'],d['src'],d['K'],d['R'],d['t'],d['rays'],d['cxcy'],d['d_candi'],8.0,feat_dist=metric,algo=algo).cpu().numpy(); fb+=_native.fallback_tiles(B,H,W)
    cd=ops.sweep_cost(d['ref'],d['src'],d['K'],d['R'],d['t'],d['rays'],d['cxcy'],d['d_candi'],8.0,feat_dist=metric,algo='direct').cpu().numpy()
    if not np.array_equal(np.isnan(ca),np.isnan(cd)): print('NaN pattern differs',case,pose,H,W,C,D,V,B,k,metric); continue
    fin=np.isfinite(cd)
    if fin.any():
        err=float(np.abs(ca-cd)[fin].max())/max(1.0,float(np.abs(cd[fin]).max())); worst=max(worst,err)
        if err>2e-6:
            print('case',case,'variant',algo,pose,H,W,C,D,V,B,k,metric,'err',err)
            if only is not None:   # replay: where?
                bad=(np.abs(ca-cd)>1e-5*max(1.0,float(np.abs(cd[fin]).max())))&fin
                bb,kk,yy,xx=np.nonzero(bad)
                print('  bad elements',bad.sum(),'of',bad.size,'planes',np.unique(kk)[:40],'rows',yy.min(),yy.max(),'cols',xx.min(),xx.max())
                tiles=sorted(set(zip((yy//4).tolist(),(xx//16).tolist())))
                print('  16x4 tiles touched',len(tiles),tiles[:24])
                for (ty,tx) in tiles[:3]:
                    sub=bad[0,:,ty*4:ty*4+4,tx*16:tx*16+16]
                    print('   tile',ty,tx,'bad per plane',sub.reshape(sub.shape[0],-1).sum(1).tolist())
    n+=1
print('cases',n,'worst',worst,'fallback tiles',fb,('worst depth difference (scaled) %.3e'%worst_d) if DPV else '')
