import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
lib=_lib.load(); dev=_lib.require_gpu()
B,d=4096,784
def direct(x0,x1):
    M=torch.empty((x0.shape[0],x1.shape[0]),dtype=torch.float32,device=dev)
    _lib.check(lib.cfm_sqeuclid_cost_f32(_lib.ptr(x0),_lib.ptr(x1),x0.shape[0],x1.shape[0],x0.shape[1],_lib.ptr(M),None,_lib.stream_ptr()),"c")
    return M
def solve(M):
    best=1e9
    for _ in range(2):
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); perm,info=ot.assign_exact(M,return_info=True); e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1))
    return best,info
res={"direct":[], "gram":[]}
for seed in (2000,3000,4000,5000,6000):
    pool=bench.synth_batches(B,d,8,seed,dev)
    for (x0,x1) in pool:
        for nm,M in (("direct",direct(x0,x1)),("gram",ot.cost_matrix(x0,x1))):
            ms,info=solve(M); res[nm].append((ms,info["stats"][3],info["stats"][2],info["stats"][6]))
for nm,v in res.items():
    a=np.array(v,dtype=float)
    print(f"{nm:7s} n={len(a)} ms {a[:,0].mean():.3f} +- {a[:,0].std()/np.sqrt(len(a)):.3f}  sapb {a[:,1].mean():.1f} +- {a[:,1].std()/np.sqrt(len(a)):.1f}  free {a[:,2].mean():.1f} steps {a[:,3].mean():.1f}")
d_=np.array(res["gram"])[:,0]-np.array(res["direct"])[:,0]
print("paired diff gram-direct ms", d_.mean(), "+-", d_.std()/np.sqrt(len(d_)))
