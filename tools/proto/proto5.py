# gap-pruning experiment: eps-scaling auction -> complete eps-optimal assignment -> G -> kept edges
import numpy as np, time, sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from proto import auction_phase
from scipy.optimize import linear_sum_assignment as lsa

def bench_batch(B,d,seed,k=0):
    g = torch.Generator().manual_seed(seed)
    mu = torch.rand(10, d, generator=g) * 2 - 1
    for _ in range(k+1):
        x0 = torch.randn(B, d, generator=g)
        kk = torch.randint(0, 10, (B,), generator=g)
        x1 = torch.clamp(0.35 * torch.randn(B, d, generator=g) + mu[kk], -1, 1)
    return x0,x1

def cost32(x0,x1):
    a=x0.double().numpy(); b=x1.double().numpy()
    M=(a*a).sum(1)[:,None]+(b*b).sum(1)[None,:]-2*a@b.T
    return np.maximum(M,0).astype(np.float32)

if __name__=="__main__":
    n=int(sys.argv[1]) if len(sys.argv)>1 else 4096
    kb=int(sys.argv[2]) if len(sys.argv)>2 else 0
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    C=M.astype(np.float64); Cr=C.max()-C.min()
    t=time.time(); r,cref=lsa(C); print("scipy",time.time()-t, "range",Cr)
    opt=C[r,cref].sum()
    theta=5.0
    p=np.zeros(n); eps=0.2*Cr
    tot=0
    for ph in range(20):
        a=np.full(n,-1); owner=np.full(n,-1); st=[]
        # cut at 2% first, record, then continue to completion
        r1,b1=auction_phase(C,p,a,owner,eps,100000,int(0.02*n),st)
        pc=p.copy(); ac=a.copy(); oc=owner.copy()
        r2,b2=auction_phase(C,pc,ac,oc,eps,100000,0,st)
        u=(C+pc[None,:]).min(1)
        sl=C[np.arange(n),ac]+pc[ac]-u
        G=sl.sum()
        Rc=C+pc[None,:]-u[:,None]
        kept=(Rc<=G*(1+1e-9)).sum(1)
        mism=(ac!=cref).sum()
        print(f"phase {ph} eps/Cr={eps/Cr:.2e} rounds_to_2%={r1} bids={b1} tail_rounds={r2} tail_bids={b2} G/Cr={G/Cr:.3e} (n*eps/Cr={n*eps/Cr:.2e}) kept/row mean={kept.mean():.1f} max={kept.max()} rows>1={np.sum(kept>1)} mism={mism} cost-opt={C[np.arange(n),ac].sum()-opt:.3e}",flush=True)
        tot+=r1
        if eps/Cr<1e-11: break
        eps/=theta
