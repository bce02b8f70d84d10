import sys, os, time, subprocess
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
from cfm_amd.ode import NeuralODE
from cfm_amd.utils import torch_wrapper
lib=_lib.load(); dev=_lib.require_gpu()
out={}
for (B,d,w,n_t) in [(8192,50,64,100),(1000,2,64,100),(333,7,48,37),(64,63,64,5)]:
    torch.manual_seed(0)
    model=cfm_amd.MLP(dim=d,time_varying=True,w=w).to(dev)
    node=NeuralODE(torch_wrapper(model),solver="dopri5",sensitivity="adjoint",atol=1e-4,rtol=1e-4)
    x=torch.randn(B,d,device=dev); ts=torch.linspace(0,1,n_t,device=dev)
    with torch.no_grad():
        traj=node.trajectory(x,ts); torch.cuda.synchronize()
        t0=time.perf_counter(); traj=node.trajectory(x,ts); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"B={B} d={d} w={w} n_t={n_t}: {dt*1e3:.2f} ms nfe={node.nfe} steps={node.n_steps} checksum={float(traj.double().sum()):.10f} last={traj[-1,0,:3].tolist()}",flush=True)
    np.save(f"/tmp/traj_{os.environ.get('CFM_ODE_FUSED','1')}_{B}_{d}.npy",traj.cpu().numpy())
    node=NeuralODE(torch_wrapper(model),solver="euler")
    with torch.no_grad():
        te=node.trajectory(x,ts); torch.cuda.synchronize()
        t0=time.perf_counter(); te=node.trajectory(x,ts); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(f"   euler: {dt*1e3:.2f} ms nfe={node.nfe} checksum={float(te.double().sum()):.10f}",flush=True)
    np.save(f"/tmp/trajE_{os.environ.get('CFM_ODE_FUSED','1')}_{B}_{d}.npy",te.cpu().numpy())
