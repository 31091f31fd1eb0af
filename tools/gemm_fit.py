#!/usr/bin/env python3
"""Fits the launch-time model of r3d_gemm_plan (r3d_amd/csrc/gemm_f32.hip) to gpurun_out/gemm_sweep.json
(written by tools/gemm_sweep.py grid on the GPU box).  Prints the constants and the regret of the model's picks."""
import json, numpy as np, re
from scipy.optimize import least_squares
import os
d=json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "gemm_sweep.json")))
TS={1:32,2:64,3:128,4:64,5:128}
rows=[]
def shape_of(name):
    m=re.match(r"(\w+?)_N(\d+)_H(\d+)",name); kind,N,H=m.group(1),int(m.group(2)),int(m.group(3))
    return {"dfwd":(0,N,H,50176),"dwg":(2,H,50176,N),"rgbf":(0,N,H,2048),"rgbwg":(2,H,2048,N),"fc1":(0,2*N,4*H,H),
            "fc2":(0,2*N,H,4*H),"dfc2":(1,2*N,4*H,H),"dfc1":(1,2*N,H,4*H),"wgfc1":(2,4*H,H,2*N)}[kind]
for name,res in d.items():
    lay,M,N,K=shape_of(name)
    for r in res[1:]:
        t,sk,us=r
        kps=-(-(-(-K//sk))//64)*64 if sk>1 else K
        ns=-(-K//kps)
        rows.append((lay,M,N,K,t,ns,kps,us,name))
print(len(rows))
def feats(lay,M,N,K,t,ns,kps):
    ts=TS[t]; tiles=(-(-M//ts))*(-(-N//ts)); wgs=tiles*ns
    ncu=-(-wgs//256); steps=-(-kps//64)
    return tiles,wgs,ncu,steps
def model(p,rows):
    out=[]
    for (lay,M,N,K,t,ns,kps,us,_) in rows:
        tiles,wgs,ncu,steps=feats(lay,M,N,K,t,ns,kps)
        lat=p[t-1]; thr=p[5+t-1]; epi=p[10+t-1]; occ=[8,4,1,2,1][t-1]
        lt = 1.0 + (0.15 if lay==2 else 0.0)*p[16]
        frac=wgs/256.0
        n=max(frac,1.0) if wgs>=256 else 1.0
        # continuous: per-CU load = wgs/256 (>=1 -> throughput bound), below that latency-bound
        load = float(ncu) if ncu <= 4 else wgs/256.0
        if lay==2 and t in (3,5): lt=p[16]
        elif lay==2: lt=p[17]
        elif lay==1: lt=p[18]
        else: lt=1.0
        step=max(lat, load*thr*lt)
        rounds=-(-ncu//occ)
        T=p[15]+steps*step+rounds*epi
        out.append(T)
    return np.array(out)
us=np.array([r[7] for r in rows])
def resid(p): return np.log(model(p,rows))-np.log(us)
p0=np.array([0.5,0.7,1.5,0.7,1.5, 0.25,0.9,3.6,0.9,3.6, 1,1,2,1,2, 3.0, 1.0,1.0,1.0])
r=least_squares(resid,p0,bounds=(1e-3,50))
p=r.x; print(np.round(p,3)); e=resid(p); print("rms log err",np.sqrt((e**2).mean()), "max",np.abs(e).max())
# how good are picks: per shape choose argmin model, compare to true best
pred=model(p,rows)
by={}
for i,rw in enumerate(rows): by.setdefault(rw[8],[]).append((pred[i],rw[7],rw[4],rw[5]))
loss=[]
for k,v in by.items():
    best=min(x[1] for x in v); pick=min(v)[1]; pk=min(v)
    loss.append(pick/best)
    if pick/best>1.08: print(k, "pick",pk[2:],pick,"best",best)
print("mean regret",np.mean(loss),"max",np.max(loss))
