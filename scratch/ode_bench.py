import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
from cfm_amd.ode import NeuralODE
from cfm_amd.utils import torch_wrapper
lib=_lib.load(); dev=_lib.require_gpu()
def run(tag,B,d,w,solver,n_t,atol=1e-4,rtol=1e-4):
    torch.manual_seed(0)
    model=cfm_amd.MLP(dim=d,time_varying=True,w=w).to(dev)
    node=NeuralODE(torch_wrapper(model),solver=solver,sensitivity="adjoint",atol=atol,rtol=rtol)
    x=torch.randn(B,d,device=dev); ts=torch.linspace(0,1,n_t,device=dev)
    with torch.no_grad():
        node.trajectory(x,ts); torch.cuda.synchronize()
        t0=time.perf_counter(); reps=5
        for _ in range(reps): traj=node.trajectory(x,ts)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    nfe=node.nfe
    flops=2*B*(((d+1)*w)+w*w*2+w*d)*nfe
    print(f"{tag}: B={B} d={d} w={w} {solver} n_t={n_t}: {dt*1e3:.2f} ms, nfe={nfe}, steps={node.n_steps}, {flops/dt/1e12:.2f} TFLOP/s MLP, {B/dt:.0f} samples/s",flush=True)
    # the same through eager torch for reference (dopri5 host loop over the tensor-level field)
run("C5 sampling",8192,50,64,"dopri5",100)
run("C5 sampling",8192,50,64,"euler",100)
run("C3-shaped",4096,784,512,"dopri5",2)
run("C3-shaped",4096,784,512,"euler",100)
run("tutorial 2-D",1024,2,64,"dopri5",100)
