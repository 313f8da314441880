"""Numerics experiment (CPU, numpy): the correlation form of the L2 sweep cost vs the fp64 oracle.

cost = w^T G w - 2 w.X + |r|^2  with G = Gram terms of the source cell, X = <ref pixel, source texel>.
Run in the build container: python tests/experiments/correlation_form_numerics.py  (results quoted in DESIGN.md 3).
"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pdepth_amd
from pdepth_amd import synth
from oracle import ref_cpu as O, c_ref
f32=np.float32
def run(pose, peaked, scale=1.0, H=64, W=128, C=67, D=64, seed=5):
    it=synth.make_item(seed,C=C,D=D,H=H,W=W,V=1,pose=pose,peaked=peaked)
    it['ref']=it['ref']*scale; it['src']=it['src']*scale
    K=it['K']; cx,cy=K.numpy()[0,2],K.numpy()[1,2]
    cost,logp,depth=O.sweep_dpv(it['ref'][None],it['src'][None],it['d_candi'],it['R'],it['t'],K,it['rays'],cx,cy,10.0)
    c64,l64,d64=c_ref.sweep_dpv_f64(it['ref'].numpy(),it['src'].numpy(),K.numpy(),it['R'].numpy(),it['t'].numpy(),it['rays'].numpy(),cx,cy,it['d_candi'],10.0)
    ix,iy=O.sample_coords(K,it['R'][0],it['t'][0],it['rays'],it['d_candi'],cx,cy,H,W)
    ix=ix.numpy().reshape(D,H,W); iy=iy.numpy().reshape(D,H,W)
    x0=np.floor(ix); y0=np.floor(iy)
    wx=(ix-x0).astype(f32); ex=(f32(1)-wx).astype(f32); ny=(iy-y0).astype(f32); sy=(f32(1)-ny).astype(f32)
    x0=np.clip(x0,-2,W+1).astype(np.int64); y0=np.clip(y0,-2,H+1).astype(np.int64)
    # zero padded src (pad 2 on low side, 3 on high side) so all accesses valid
    P=3
    src=np.zeros((C,H+2*P,W+2*P),f32); src[:,P:P+H,P:P+W]=it['src'][0].numpy()
    ref=it['ref'].numpy()
    dot=lambda a,b: np.einsum('chw,chw->hw',a,b,dtype=f32)
    S=src
    N=dot(S,S)
    Hh=np.zeros_like(N); Hh[:,:-1]=dot(S[:,:,:-1],S[:,:,1:])
    Vv=np.zeros_like(N); Vv[:-1,:]=dot(S[:,:-1,:],S[:,1:,:])
    D1=np.zeros_like(N); D1[:-1,:-1]=dot(S[:,:-1,:-1],S[:,1:,1:])
    D2=np.zeros_like(N); D2[:-1,:-1]=dot(S[:,:-1,1:],S[:,1:,:-1])
    rr=np.einsum('chw,chw->hw',ref,ref,dtype=f32)
    yy,xx=np.meshgrid(np.arange(H),np.arange(W),indexing='ij')
    costd=np.zeros((D,H,W),f32)
    for k in range(D):
        X0=x0[k]+P; Y0=y0[k]+P
        def X(dx,dy):
            s=S[:,Y0+dy,X0+dx]  # [C,H,W]
            return np.einsum('chw,chw->hw',s,ref,dtype=f32)
        Xa,Xb,Xc,Xd=X(0,0),X(1,0),X(0,1),X(1,1)
        G00,G11,G22,G33=N[Y0,X0],N[Y0,X0+1],N[Y0+1,X0],N[Y0+1,X0+1]
        G01,G23=Hh[Y0,X0],Hh[Y0+1,X0]; G02,G13=Vv[Y0,X0],Vv[Y0,X0+1]; G03=D1[Y0,X0]; G12=D2[Y0,X0]
        e,w,n,s_=ex[k],wx[k],ny[k],sy[k]
        ee=(e*e).astype(f32); ww=(w*w).astype(f32); ew=(e*w).astype(f32)
        A=(ee*G00+ww*G11+f32(2)*ew*G01).astype(f32)
        Bq=(ee*G22+ww*G33+f32(2)*ew*G23).astype(f32)
        Cq=(ee*G02+ww*G13+ew*(G03+G12)).astype(f32)
        Q=((s_*s_)*A+(n*n)*Bq+f32(2)*(s_*n)*Cq).astype(f32)
        nw,ne,sw,se=(s_*e).astype(f32),(s_*w).astype(f32),(n*e).astype(f32),(n*w).astype(f32)
        XW=(nw*Xa+ne*Xb+sw*Xc+se*Xd).astype(f32)
        costd[k]=((Q-f32(2)*XW+rr)/f32(10)).astype(f32)
    # softmax/expect in f64 from fp32 costs to isolate cost error
    def dep(c):
        c=c.astype(np.float64); m=c.max(0); p=np.exp(c-m); p/=p.sum(0); return (p*it['d_candi'].astype(f32).astype(np.float64)[:,None,None]).sum(0)
    print(f"{pose:6s} peaked={peaked} scale={scale}: cost range [{c64.min():.2f},{c64.max():.2f}]",
          f"| oracle-f64 cost {np.abs(cost.numpy()[0]-c64).max():.2e} depth {np.abs(depth.numpy()[0]-d64).max():.2e}",
          f"| decomp-f64 cost {np.abs(costd-c64).max():.2e} depth {np.abs(dep(costd)-d64).max():.2e}",
          f"| decomp-oracle depth {np.abs(dep(costd)-depth.numpy()[0]).max():.2e}")
for pose in ('mono','stereo'):
    for peaked in (False,True):
        run(pose,peaked)
run('mono',False,scale=3.0)
run('mono',True,scale=0.3)
