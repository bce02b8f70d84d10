import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
from cfm_amd import _lib
_lib.LIB_PATH=os.path.join(ROOT,'scratch','variants','bidprof.so')
import cfm_amd.optimal_transport as ot
from cfm_amd._lib import ptr, stream_ptr
import bench
lib=_lib.load(); dev=_lib.require_gpu()
x0,x1=bench.synth_batches(4096,784,1,1000,dev)[0]
M=ot.cost_matrix(x0,x1); B=4096
for rep in range(2):
    perm=torch.empty(B,dtype=torch.int32,device=dev); cert=torch.zeros(1,dtype=torch.int32,device=dev)
    tot=torch.zeros(1,dtype=torch.float64,device=dev); stats=torch.zeros(8,dtype=torch.int32,device=dev)
    ws=_lib.workspace(_lib.OP_ASSIGN,B,B,0,dev)
    lib.cfm_assign_exact_f32(ptr(M),B,ptr(perm),ptr(cert),ptr(tot),ptr(stats),ptr(ws),stream_ptr())
    torch.cuda.synchronize()
    d=ws[400:456].cpu().view(torch.int64).tolist()
    n=max(d[6],1)
    print("small bid rounds:",d[6]," avg ticks [entry->bid fn, issue loads+lds write, barrier, wait row, compute, commit]:",[round(x/n) for x in d[:6]])
