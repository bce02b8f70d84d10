import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
if os.environ.get('CFM_LIB_OVERRIDE'): _lib.LIB_PATH=os.environ['CFM_LIB_OVERRIDE']
lib=_lib.load(); dev=_lib.require_gpu()
tag=os.environ.get('CFM_SK_FUSED','1')
def pots(r,B0,B1):
    u=torch.empty(B0,dtype=torch.float64,device=dev); v=torch.empty(B1,dtype=torch.float64,device=dev)
    _lib.check(lib.cfm_sinkhorn_potentials_f64(_lib.ptr(r.ws),B0,B1,_lib.ptr(u),_lib.ptr(v),_lib.stream_ptr()),"p")
    return u.cpu().numpy(),v.cpu().numpy()
for name,B,reg,iters in (("C2",4096,0.05,200),("C2",4096,0.05,35),("C2b",2048,0.5,1000),("rect",None,1.0,60)):
    if name=="rect":
        g=torch.Generator().manual_seed(1); x0=torch.randn(1500,3,generator=g); x1=torch.randn(1024,3,generator=g)+0.5
    else:
        x0,x1=oracle.config_inputs("C2",B=B)
    M=ot.cost_matrix(x0.to(dev),x1.to(dev))
    stop=0.0 if iters!=1000 else 1e-9
    r=ot.sinkhorn_log(M,reg,max_iter=iters,stop_thr=stop); torch.cuda.synchronize()
    t0=time.perf_counter(); r=ot.sinkhorn_log(M,reg,max_iter=iters,stop_thr=stop); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    u,v=pots(r,*M.shape); it=int(r.iters.item())
    print(f"[{tag}] {name} {tuple(M.shape)} reg={reg} max_iter={iters}: {dt*1e3:.2f} ms, iters={it}, {dt/max(it,1)*1e6:.1f} us/iter, err={float(r.err.item()):.3e}, u[0]={u[0]:.12f} v[0]={v[0]:.12f} |u|max={np.abs(u).max():.6f}",flush=True)
    np.save(f"/tmp/sk_{tag}_{name}_{iters}.npy",np.concatenate([u,v]))
    if iters<=60:
        uo,vo,_,_=oracle.sinkhorn_log(M.cpu().numpy(),reg,numItermax=iters,stopThr=0.0)
        print(f"      vs oracle: rel err u {np.abs(u-uo).max()/np.abs(uo).max():.2e}, v {np.abs(v-vo).max()/np.abs(vo).max():.2e}")
