import sys, os
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+"/oracle")
import numpy as np, torch, cfm_amd, cfm_oracle as oracle
from cfm_amd import _lib
_lib.load(); dev=_lib.require_gpu()
def rel(a,b): return float(np.abs(np.asarray(a,dtype=np.float64)-b).max()/np.abs(b).max())
for B,d,w in ((512,784,512),(4096,784,512),(300,50,64)):
    torch.manual_seed(B+d)
    m=cfm_amd.MLP(dim=d,time_varying=True,w=w).to(dev)
    g=torch.Generator().manual_seed(1); x=torch.randn(B,d+1,generator=g); ut=torch.randn(B,d,generator=g)
    res={}
    for hip in (True,False):
        m.hip_training=hip; m.zero_grad(set_to_none=True)
        xin=x.to(dev).requires_grad_(True)
        vt=m(xin); loss=torch.mean((vt-ut.to(dev))**2); loss.backward()
        res[hip]=(vt.detach().cpu().numpy(), [l.weight.grad.cpu().numpy() for l in m._linears()], [l.bias.grad.cpu().numpy() for l in m._linears()], xin.grad.cpu().numpy())
    Ws=[l.weight.detach().cpu().numpy() for l in m._linears()]; bs=[l.bias.detach().cpu().numpy() for l in m._linears()]
    out_o=oracle.mlp_forward_f64(Ws,bs,x.numpy()); dout=2.0*(out_o-ut.numpy().astype(np.float64))/(B*d)
    _,dW,db,dx=oracle.mlp_backward_f64(Ws,bs,x.numpy(),dout)
    # oracle backward fed with the fp32 forward's own dout (isolates the backward kernels from the forward error)
    for hip in (True,False):
        o,gw,gb,gx=res[hip]
        print(B,d,w,"HIP" if hip else "torch","out",f"{rel(o,out_o):.2e}","dW",[f"{rel(a,b):.1e}" for a,b in zip(gw,dW)],"db",[f"{rel(a,b):.1e}" for a,b in zip(gb,db)],"dx",f"{rel(gx,dx):.1e}",flush=True)
    dout32=2.0*(res[True][0].astype(np.float64)-ut.numpy().astype(np.float64))/(B*d)
    _,dW2,db2,dx2=oracle.mlp_backward_f64(Ws,bs,x.numpy(),dout32.astype(np.float32))
    o,gw,gb,gx=res[True]
    print("   HIP backward vs oracle backward on the SAME fp32 dout: dW",[f"{rel(a,b):.1e}" for a,b in zip(gw,dW2)],"dx",f"{rel(gx,dx2):.1e}")
