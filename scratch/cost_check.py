import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
def direct(x0,x1):
    M=torch.empty((x0.shape[0],x1.shape[0]),dtype=torch.float32,device=dev)
    _lib.check(lib.cfm_sqeuclid_cost_f32(_lib.ptr(x0),_lib.ptr(x1),x0.shape[0],x1.shape[0],x0.shape[1],_lib.ptr(M),None,_lib.stream_ptr()),"c")
    return M
def stats(name,x0,x1):
    a=x0.to(dev); b=x1.to(dev)
    Mg=ot.cost_matrix(a,b).cpu().numpy().astype(np.float64); Md=direct(a,b).cpu().numpy().astype(np.float64)
    ref=oracle.sqeuclid_cost_f64(x0.numpy(),x1.numpy())
    den=np.maximum(ref,1e-30)
    eg=np.abs(Mg-ref)/den; ed=np.abs(Md-ref)/den
    nz=ref>0
    print(f"{name:34s} d={x0.shape[1]:4d} gram: max rel {eg[nz].max():.2e} (99.9% {np.quantile(eg[nz],0.999):.2e})  direct: max rel {ed[nz].max():.2e}  | zeros exact: {bool((Mg[~nz]==0).all())} min {Mg.min():.3g}",flush=True)
g=torch.Generator().manual_seed(0)
def rn(*s): return torch.randn(*s,generator=g)
stats("randn vs randn+0.3",rn(512,784),rn(512,784)+0.3)
stats("randn vs randn, d=64",rn(300,64),rn(260,64))
stats("d=65 (scalar loads)",rn(257,65),rn(515,65))
stats("d=100",rn(1030,100),rn(1100,100))
stats("offset +100",rn(512,784)+100,rn(512,784)+100.3)
stats("offset +1e4",rn(512,128)+1e4,rn(512,128)+1e4)
x=rn(512,784); stats("x vs x",x,x.clone())
stats("near duplicates 1e-3",x,x+1e-3*rn(512,784))
stats("clusters (8 tight blobs)",(rn(8,200)*5)[torch.arange(512)%8]+0.01*rn(512,200),(rn(8,200)*5)[torch.arange(640)%8]+0.01*rn(640,200))
stats("mnist-like (noise vs [0,1] px)",rn(1024,784),torch.rand(1024,784,generator=g).pow(4))
stats("scaled 1e-3",rn(512,784)*1e-3,rn(512,784)*1e-3)
stats("sorted drift",torch.sort(rn(2048,96)+torch.linspace(0,50,2048)[:,None],0)[0],rn(2048,96)+25)
# timing
for B,d in ((4096,784),(4096,128),(8192,784),(2048,64)):
    a=rn(B,d).to(dev); b=(rn(B,d)+0.3).to(dev)
    for nm,f in (("gram/mfma",lambda: ot.cost_matrix(a,b)),("direct",lambda: direct(a,b))):
        f(); torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(f"B={B} d={d} {nm:10s} {ms*1e3:8.1f} us  {2*B*B*d/ms/1e9:7.1f} TFLOP/s(2 flop/elem-k)",flush=True)
# degenerate: all points identical -> everything recomputed
a=torch.ones(2048,256).to(dev)
t0=time.perf_counter(); M=ot.cost_matrix(a,a.clone()); torch.cuda.synchronize(); print("all-identical 2048x2048x256:",f"{(time.perf_counter()-t0)*1e3:.1f} ms, max",float(M.max()))
