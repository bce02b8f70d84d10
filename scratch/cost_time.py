import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
g=torch.Generator().manual_seed(0)
def rn(*s): return torch.randn(*s,generator=g)
x0=rn(1024,784); x1=rn(1024,784)+0.3
M=ot.cost_matrix(x0.to(dev),x1.to(dev)).cpu().numpy().astype(np.float64); ref=oracle.sqeuclid_cost_f64(x0.numpy(),x1.numpy())
print("1024 max rel", (np.abs(M-ref)/ref).max())
x0=rn(4096,784); x1=rn(4096,784)+0.3
M=ot.cost_matrix(x0.to(dev),x1.to(dev)).cpu().numpy().astype(np.float64); ref=oracle.sqeuclid_cost_f64(x0.numpy(),x1.numpy())
print("4096 max rel", (np.abs(M-ref)/ref).max())
for B,d in ((4096,784),(8192,784),(4096,128),(4000,784)):
    a=rn(B,d).to(dev); b=(rn(B,d)+0.3).to(dev)
    f=lambda: ot.cost_matrix(a,b)
    f(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print(f"B={B} d={d} {ms*1e3:8.1f} us  {2*B*B*d/ms/1e9:7.1f} TFLOP/s",flush=True)
