import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
from cfm_amd.conditional_flow_matching import ExactOptimalTransportConditionalFlowMatcher
import bench
lib=_lib.load(); dev=_lib.require_gpu()
B,d=4096,784
pool=bench.synth_batches(B,d,8,1000,dev)
fm=ExactOptimalTransportConditionalFlowMatcher(sigma=0.0)
torch.manual_seed(0)
model=cfm_amd.MLP(dim=d,time_varying=True,w=512).to(dev)
opt=torch.optim.Adam(model.parameters(),lr=1e-3)
names=["cost","assign","u01+sample","xt_ut","model fwd","loss+bwd","adam"]
acc=np.zeros(len(names)); wall=np.zeros(len(names)); n=0
def mark(): 
    e=torch.cuda.Event(enable_timing=True); e.record(); return e, time.perf_counter()
for k in range(24):
    x0,x1=pool[k%8]
    torch.cuda.synchronize()
    ev=[mark()]
    M=ot.cost_matrix(x0,x1); ev.append(mark())
    perm,info=ot.assign_exact(M,return_info=True); ev.append(mark())
    u=torch.from_numpy(np.random.random_sample(B)).to(dev); i,j=ot.sample_perm(perm,u,B); ev.append(mark())
    t,xt,ut=fm._sample(x0,x1,None,False,idx=(i,j)); ev.append(mark())
    opt.zero_grad(set_to_none=True); vt=model(torch.cat([xt,t[:,None]],dim=-1)); ev.append(mark())
    loss=torch.mean((vt-ut)**2); loss.backward(); ev.append(mark())
    opt.step(); ev.append(mark())
    torch.cuda.synchronize(); tend=time.perf_counter()
    if k>=4:
        for q in range(len(names)):
            acc[q]+=ev[q][0].elapsed_time(ev[q+1][0]); wall[q]+=(ev[q+1][1]-ev[q][1])*1e3
        wall[-1]+=0; n+=1
print("stage: gpu-event ms | host-wall ms (launch side)")
for q,nm in enumerate(names): print(f"  {nm:12s} {acc[q]/n:7.3f} | {wall[q]/n:7.3f}")
print("  total gpu-event", acc.sum()/n, " host", wall.sum()/n)
