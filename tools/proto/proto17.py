# Tail of the exact assignment (<= 6 free rows): sequential single-source searches (what sp_solver
# does) vs multi-source forest phases (what the wide phases do, here carried to the end).
# Counts label-correcting rounds and row relaxations ("scans": one list entry of a solver batch).
import numpy as np, sys, copy
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from scipy.optimize import linear_sum_assignment as lsa

def forest_phase(C,p,a,owner,roots,cap=64):
    """One pruned label-correcting forest from `roots`; accepts one path per tree that reached a free
    column below the radius.  Returns (augmented, rounds, scans)."""
    n=C.shape[0]
    roots=np.asarray(roots); freec=owner<0
    V=C[roots]+p[None,:]; u0=V.min(1)
    R0=np.maximum(V-u0[:,None],0.0)
    src=R0.argmin(0); d=R0[src,np.arange(n)]
    pred=roots[src].copy(); root=roots[src].copy()
    dirty=(owner>=0)
    rounds=0; scans=0
    def radius():
        fcs=np.where(freec)[0]
        best={}
        for k in fcs:
            r=root[k]
            if d[k]<best.get(r,np.inf): best[r]=d[k]
        if len(best)==len(roots): return max(best.values())
        if len(roots)==1 and len(best)==1: return list(best.values())[0]
        return np.inf
    while True:
        D=radius()
        S=np.where(dirty&(d<D))[0]
        if len(S)==0: break
        # a solver batch takes at most 64 entries: lowest labels first (delta-stepping-like)
        if len(S)>cap:
            S=S[np.argsort(d[S],kind='stable')[:cap]]
        dirty[S]=False
        rows=owner[S]
        rj=C[rows,S]+p[S]
        RC=np.maximum(C[rows]+p[None,:]-rj[:,None],0.0)
        cand=d[S][:,None]+RC
        cand[np.arange(len(S)),S]=np.inf
        m=cand.argmin(0); cm=cand[m,np.arange(n)]
        upd=cm<d
        d[upd]=cm[upd]; pred[upd]=rows[m[upd]]; root[upd]=root[S[m[upd]]]
        dirty[upd&(owner>=0)]=True
        rounds+=1; scans+=len(S)
    def true_root(k):
        i=pred[k]; g=0
        while a[i]>=0:
            i=pred[a[i]]; g+=1
            if g>n: raise RuntimeError
        return i
    fcs=np.where(freec)[0]
    best={}
    for k in fcs:
        if not np.isfinite(d[k]): continue
        r=true_root(k)
        if r not in best or d[k]<d[best[r]]: best[r]=k
    D=max(d[k] for k in best.values())
    best={r:k for r,k in best.items() if d[k]<=D}
    inT=(d<D); p[inT]+=D-d[inT]
    for r,k in best.items():
        j=k
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j
            if i==r: break
            j=jprev
    return len(best),rounds,scans

def handoff_state(M,handoff=6):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    u=C.min(1); p=-(C-u[:,None]).min(0)
    a=np.full(n,-1); owner=np.full(n,-1); eps=Cr*8e-3; stats=[]
    while eps>=Cr*1e-6:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(0.02*n),stats)
        eps/=5
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,15,0,st)
    col_reduce(C,p,owner)
    ms=[]
    while np.sum(a<0)>handoff:
        k,r,s=forest_phase(C,p,a,owner,np.where(a<0)[0]); ms.append((k,r,s))
    return C,p,a,owner,ms

if __name__=="__main__":
    n=int(sys.argv[1]) if len(sys.argv)>1 else 4096
    caps=(64,128,256,100000)
    tot={c:[0,0] for c in caps}; totms=[0,0]
    for kb in range(int(sys.argv[2]) if len(sys.argv)>2 else 4):
        x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
        C,p,a,owner,ms=handoff_state(M)
        print(f"inst {kb}: wide MS phases {ms}  tail free {int((a<0).sum())}")
        for cap in caps:
            p2,a2,o2=p.copy(),a.copy(),owner.copy(); R=0; S=0; phases=[]
            while (a2<0).any():
                fr=np.where(a2<0)[0]
                k,r,s=forest_phase(C,p2,a2,o2,fr[:1],cap); R+=r; S+=s; phases.append((r,s))
            tot[cap][0]+=R; tot[cap][1]+=S
            print(f"    seq cap {cap:6d}: rounds {R:4d} scans {S:6d}  per search (rounds,scans) {phases}",flush=True)
    print("total (rounds, scans) per cap:",tot)
