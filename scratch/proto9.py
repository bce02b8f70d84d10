# multi-source (Voronoi forest) SAP: one label-correcting pass from ALL free rows, augment one
# path per tree that reached a free column; count phases / BF rounds / row relaxations.
import numpy as np, time, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from scipy.optimize import linear_sum_assignment as lsa

def multi_phase(C,p,a,owner,mode):
    n=C.shape[0]
    free_rows=np.where(a<0)[0]; freec=owner<0
    V=C[free_rows]+p[None,:]
    u0=V.min(1)
    R0=np.maximum(V-u0[:,None],0.0)              # [F,n]
    src=R0.argmin(0); d=R0[src,np.arange(n)]
    pred=free_rows[src].copy()                   # predecessor row
    root=free_rows[src].copy()                   # tree id (root row)
    dirty=(owner>=0)
    rounds=0; scans=0
    while True:
        # pruning radius
        D=np.inf
        if mode=="prune":
            fcs=np.where(freec)[0]
            best={}
            for k in fcs:
                r=root[k]
                if d[k]<best.get(r,np.inf): best[r]=d[k]
            if len(best)==len(free_rows): D=max(best.values())
        S=np.where(dirty&(d<D))[0]
        if len(S)==0: break
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
    # root ids may be stale for descendants if an ancestor's root changed later; recompute by walking pred
    # pick nearest free column per tree (walk to the root to find the true tree)
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
    # dual update radius D
    inT=(d<D)
    p[inT]+=D-d[inT]       # includes free columns with label < D
    # augment each chosen path
    for r,k in best.items():
        j=k
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j
            if i==r: break
            j=jprev
    return len(best),rounds,scans,D

def prep(M):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*0.2; stats=[]
    while eps>=Cr*1e-6:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(0.02*n),stats)
        col_reduce(C,p,owner)
        eps/=5
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,30,0,st)
    col_reduce(C,p,owner)
    return C,Cr,p,a,owner

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2])
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    r,cref=lsa(M.astype(np.float64))
    C,Cr,p0,a0,o0=prep(M)
    for mode in ["full","prune"]:
        p=p0.copy(); a=a0.copy(); owner=o0.copy()
        print("mode",mode,"free",np.sum(a<0))
        tr=0; ts=0; ph=0
        while np.any(a<0):
            k,rounds,scans,D=multi_phase(C,p,a,owner,mode)
            print(f"   phase {ph}: augmented {k} rounds={rounds} scans={scans} D/Cr={D/Cr:.3e}")
            tr+=rounds; ts+=scans; ph+=1
        ar=np.arange(n); u=C[ar,a]+p[a]; S=C+p[None,:]-u[:,None]
        print(f"   total phases={ph} rounds={tr} scans={ts} mism={(a!=cref).sum()} minslack/Cr={S.min()/Cr:.2e}")
