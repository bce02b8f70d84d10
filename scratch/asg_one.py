import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
import os
if os.environ.get('CFM_LIB_OVERRIDE'): _lib.LIB_PATH=os.environ['CFM_LIB_OVERRIDE']
lib=_lib.load(); dev=_lib.require_gpu()
name=sys.argv[1] if len(sys.argv)>1 else "C3"
reps=int(sys.argv[2]) if len(sys.argv)>2 else 3
x0,x1=oracle.config_inputs(name,B=4096)
M=ot.cost_matrix(x0.to(dev),x1.to(dev))
for r in range(reps):
    torch.cuda.synchronize(); t0=time.perf_counter()
    perm,info=ot.assign_exact(M,return_info=True)
    torch.cuda.synchronize(); print(f"{name} rep{r} {1e3*(time.perf_counter()-t0):.2f} ms {info['stats']}",flush=True)
if not os.environ.get('CFM_PROFILE'): sys.exit(0)
# profile dump (only meaningful when built with -DSP_PROFILE)
import ctypes
from cfm_amd._lib import ptr, stream_ptr
B=M.shape[0]
perm=torch.empty(B,dtype=torch.int32,device=dev); cert=torch.zeros(1,dtype=torch.int32,device=dev)
tot=torch.zeros(1,dtype=torch.float64,device=dev); stats=torch.zeros(8,dtype=torch.int32,device=dev)
ws=_lib.workspace(_lib.OP_ASSIGN,B,B,0,dev)
lib.cfm_assign_exact_f32(ptr(M),B,ptr(perm),ptr(cert),ptr(tot),ptr(stats),ptr(ws),stream_ptr())
torch.cuda.synchronize()
d=ws[256:384].cpu().view(torch.int64).tolist()
print("solver profile cycles [init,fast_relax,collect,check,finish,generic_relax,n_fast,batches]:",d)
nb=max(d[7],1); nf=max(d[6],1); ng=max(d[7]-d[6],1)
print("  matched-col list misses:",d[8]," avg near-list length per collect:",d[15]/max(d[7],1)); print("  fast-batch per-batch cycles [loads,lower,sync1,dfree,claim,sync2]:",[round(x/max(d[6],1)) for x in d[9:15]]); print("  per fast batch relax %.0f cyc; per generic batch relax %.0f cyc; collect per batch %.0f"%(d[1]/nf,d[5]/ng,d[2]/nb))
