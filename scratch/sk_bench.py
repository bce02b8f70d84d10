import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
if os.environ.get('CFM_LIB_OVERRIDE'): _lib.LIB_PATH=os.environ['CFM_LIB_OVERRIDE']
lib=_lib.load(); dev=_lib.require_gpu()
for name,B,reg in (("C2",4096,0.05),("C5",8192,0.1)):
    x0,x1=oracle.config_inputs(name,B=B)
    M=ot.cost_matrix(x0.to(dev),x1.to(dev))
    ot.sinkhorn_log(M,reg,max_iter=20,stop_thr=0.0); torch.cuda.synchronize()
    iters=200
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); r=ot.sinkhorn_log(M,reg,max_iter=iters,stop_thr=0.0); e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1); byt=(2*4*B*B+16*B)*iters
    print(f"{name} B={B}: {iters/ms*1e3:.0f} it/s, {ms/iters*1e3:.1f} us/iter, {byt/ms/1e6:.0f} GB/s ({byt/ms/1e6/80:.1f}% of 8 TB/s)",flush=True)
