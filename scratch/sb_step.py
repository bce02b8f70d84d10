import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
from cfm_amd.conditional_flow_matching import SchrodingerBridgeConditionalFlowMatcher
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
def ev(): e=torch.cuda.Event(enable_timing=True); e.record(); return e
for name,B,reg in (("C2",4096,0.05),("C5",8192,0.1)):
    x0,x1=oracle.config_inputs(name,B=B); x0=x0.to(dev); x1=x1.to(dev)
    for rep in range(2):
        torch.cuda.synchronize(); e0=ev()
        M=ot.cost_matrix(x0,x1); e1=ev()
        r=ot.sinkhorn_log(M,reg); e2=ev()
        u=torch.from_numpy(np.random.random_sample(B)).to(dev); i,j=ot.sample_dense(r,u); e3=ev()
        torch.cuda.synchronize()
    print(f"{name} B={B} reg={reg}: cost {e0.elapsed_time(e1):.3f} ms, sinkhorn {e1.elapsed_time(e2):.2f} ms ({int(r.iters.item())} iterations, err {float(r.err.item()):.2e}), dense sampling {e2.elapsed_time(e3):.3f} ms",flush=True)
    fm=SchrodingerBridgeConditionalFlowMatcher(sigma=float(np.sqrt(reg/2)),ot_method="sinkhorn")
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(3): out=fm.sample_location_and_conditional_flow(x0,x1)
    torch.cuda.synchronize(); print(f"   SB-CFM sample_location_and_conditional_flow: {(time.perf_counter()-t0)/3*1e3:.2f} ms per call")
print("---- breakdown C2")
x0,x1=oracle.config_inputs("C2",B=4096); x0=x0.to(dev); x1=x1.to(dev)
fm=SchrodingerBridgeConditionalFlowMatcher(sigma=float(np.sqrt(0.05/2)),ot_method="sinkhorn")
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    i,j=fm.ot_sampler._sample_indices(x0,x1); torch.cuda.synchronize(); t1=time.perf_counter()
    out=fm._sample(x0,x1,None,False,idx=(i,j)); torch.cuda.synchronize(); t2=time.perf_counter()
    r=fm.ot_sampler._last
    print(f"   _sample_indices {1e3*(t1-t0):.2f} ms (iters {int(r.iters.item())}), _sample {1e3*(t2-t1):.2f} ms, reg {fm.ot_sampler.reg}")
