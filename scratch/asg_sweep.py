import sys, os, time, json
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
def run(M, reps=3):
    ts=[]; info=None
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter()
        perm,info=ot.assign_exact(M,return_info=True)
        torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return min(ts)*1e3, info['stats'], perm
cfgs={}
which=sys.argv[1].split(',') if len(sys.argv)>1 else ["C3","C2"]
for name,B in (("C3",4096),("C2",4096),("C5",8192)):
    if name not in which: continue
    x0,x1=oracle.config_inputs(name,B=B)
    cfgs[name]=ot.cost_matrix(x0.to(dev),x1.to(dev))
ref={}
params=[(5,0.2,1e-6,0.02,4000,30,48),(5,0.2,1e-6,0.01,4000,60,48),(4,0.1,1e-7,0.02,4000,30,48),(5,0.2,1e-4,0.02,4000,30,48),(5,0.2,1e-5,0.05,4000,20,48),(10,0.2,1e-5,0.05,4000,10,48)]
for sparse in (1,0):
  lib.cfm_assign_set_mode(sparse)
  for p in (params if sparse else params[:1]):
    lib.cfm_assign_set_params(*map(float,p[:4]),*map(int,p[4:]))
    for name in which:
        ms,st,perm=run(cfgs[name])
        key=name
        if key not in ref: ref[key]=perm.clone()
        same=bool(torch.equal(ref[key],perm))
        print(f"{name} sparse={sparse} params={p} -> {ms:8.2f} ms stats[auct,arr,free,batches,sapscans,scans,steps,phases]={st} same_perm={same}",flush=True)
