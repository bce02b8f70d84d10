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
pool=bench.synth_batches(B,d,8,1000,dev)
def direct(x0,x1):
    M=torch.empty((x0.shape[0],x1.shape[0]),dtype=torch.float32,device=dev)
    _lib.check(lib.cfm_sqeuclid_cost_f32(_lib.ptr(x0),_lib.ptr(x1),x0.shape[0],x1.shape[0],x0.shape[1],_lib.ptr(M),None,_lib.stream_ptr()),"c")
    return M
def solve(M):
    best=1e9
    for _ in range(3):
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); perm,info=ot.assign_exact(M,return_info=True); e1.record(); torch.cuda.synchronize()
        best=min(best,e0.elapsed_time(e1))
    return best,info,perm
tot={}
for k,(x0,x1) in enumerate(pool):
    Mg=ot.cost_matrix(x0,x1); Md=direct(x0,x1)
    # a third matrix: direct + multiplicative noise of the Gram-vs-direct size
    g=torch.Generator(device=dev).manual_seed(k)
    Mn=Md*(1+3e-7*torch.randn(Md.shape,generator=g,device=dev))
    for nm,M in (("direct",Md),("gram",Mg),("direct+noise",Mn)):
        ms,info,perm=solve(M)
        s=info["stats"]
        tot.setdefault(nm,[]).append(ms)
        print(f"inst {k} {nm:13s} {ms:6.2f} ms  rounds {s[0]:3d} arr {s[1]:2d} free {s[2]:3d} sapb {s[3]:4d} saprows {s[4]:6d} rows {s[5]:7d} steps {s[6]:4d} packed {s[7]:#x} cost {info['total_cost']:.6f}",flush=True)
    print("   perms equal gram/direct:", bool((ot.assign_exact(Mg)==ot.assign_exact(Md)).all().item()))
for nm,v in tot.items(): print(nm, "mean ms", np.mean(v))
