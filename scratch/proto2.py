import numpy as np, time, sys
sys.path.insert(0,'.')
import gen
from proto import auction_phase, sap
from scipy.optimize import linear_sum_assignment as lsa

def run(M,cref,theta,eps0_frac,eps_last_frac,frac_stop,arr_rounds,reset=True):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0_frac; stats=[]
    while eps>=Cr*eps_last_frac:
        if reset: a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(frac_stop*n),stats)
        eps/=theta
    R=sum(s[1] for s in stats); Bd=sum(s[2] for s in stats)
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    free=np.where(a<0)[0]
    v=-p; u=(C+p[None,:]).min(1)
    steps=sap(C,u,v,a,owner,free)
    mism=(a!=cref).sum()
    print(f"theta={theta} e0={eps0_frac} elast={eps_last_frac:g} stop={frac_stop} | phases={len(stats)} rounds={R} bids/n={Bd/n:.1f} | ARR rounds={st[0][1]} bids/n={st[0][2]/n:.2f} free={len(free)} | SAP steps={sum(steps)} max={max(steps) if steps else 0} | mismatch={mism}")
    return stats

if __name__=="__main__":
    cfg=sys.argv[1]; n=int(sys.argv[2])
    x0,x1=gen.get(cfg,n); M=gen.cost(x0,x1)
    t=time.time(); r,cref=lsa(M.astype(np.float64)); print("scipy",time.time()-t)
    for (theta,e0,el,fs,ar) in [(5,0.2,1e-4,0.02,30),(5,0.2,1e-6,0.02,30),(5,0.2,1e-8,0.02,30),(4,0.05,1e-6,0.05,30),(10,0.2,1e-6,0.01,30),(5,0.2,1e-6,0.005,100),(5,0.2,1e-6,0.02,0)]:
        t=time.time(); run(M,cref,theta,e0,el,fs,ar); print("   t=%.1fs"%(time.time()-t))
