# Multi-source forest phases ON CANDIDATE LISTS (K = 64 cheapest columns per row + bound T_i), with the
# a-posteriori test in its multi-source form, followed by the sequential list searches of proto4 for the
# last rows: a reference for moving the multi-source phases of phase C into the one-workgroup solver.
# Validated against SciPy's optimum; counts batches (<= 64 list entries each).
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from proto import auction_phase
from proto4 import build_lists, sap_sparse
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from scipy.optimize import linear_sum_assignment as lsa

def forest_lists_phase(C,p,a,owner,roots,cols,T,cap=64):
    n=C.shape[0]; roots=np.asarray(roots); freec=owner<0
    d=np.full(n,np.inf); pred=np.full(n,-1); root=np.full(n,-1)
    u0={}
    for r in roots:
        ck=cols[r]; val=C[r,ck]+p[ck]; u=val.min(); u0[r]=u
        c=np.maximum(val-u,0.0); better=c<d[ck]
        kk=ck[better]; d[kk]=c[better]; pred[kk]=r; root[kk]=r
    dirty=(owner>=0)&np.isfinite(d)
    dense_done=np.full(n,np.inf); root_dense=set()
    batches=0; scans=0; dense=0
    def radius():
        best={}
        for k in np.where(freec&np.isfinite(d))[0]:
            r=root[k]
            if d[k]<best.get(r,np.inf): best[r]=d[k]
        return (max(best.values()) if best else np.inf), best
    while True:
        while True:
            D,_=radius()
            S=np.where(dirty&(d<D))[0]
            if len(S)==0: break
            if len(S)>cap: S=S[np.argsort(d[S],kind='stable')[:cap]]
            dirty[S]=False
            nd=d.copy(); npred=pred.copy(); nroot=root.copy()
            for j in S:
                i=owner[j]; rj=C[i,j]+p[j]; ks=cols[i]; ks=ks[ks!=j]
                cand=d[j]+np.maximum((C[i,ks]+p[ks])-rj,0.0)
                b=cand<nd[ks]; kk=ks[b]; nd[kk]=cand[b]; npred[kk]=i; nroot[kk]=root[j]
            imp=nd<d; d=nd; pred=npred; root=nroot
            dirty|=imp&(owner>=0)
            batches+=1; scans+=len(S)
        # a-posteriori test, multi-source form: no dropped edge of a row labelled below D reaches below D
        D,_=radius()
        bad=[]
        for r in roots:
            if r not in root_dense and not (T[r]-u0[r]>=D): bad.append(('r',r))
        for j in np.where((owner>=0)&(d<D))[0]:
            i=owner[j]; rj=C[i,j]+p[j]
            if d[j]<dense_done[j] and not (d[j]+(T[i]-rj)>=D): bad.append(('c',j))
        if not bad: break
        nd=d.copy(); npred=pred.copy(); nroot=root.copy()
        for kind,x in bad:
            dense+=1
            if kind=='r':
                root_dense.add(x); cand=np.maximum((C[x]+p)-u0[x],0.0); i=x; rt=x
            else:
                i=owner[x]; rj=C[i,x]+p[x]; dense_done[x]=d[x]
                cand=d[x]+np.maximum((C[i]+p)-rj,0.0); cand[x]=np.inf; rt=root[x]
            b=cand<nd; nd[b]=cand[b]; npred[b]=i; nroot[b]=rt
        imp=nd<d; d=nd; pred=npred; root=nroot
        dirty|=imp&(owner>=0)
        batches+=1
    # accept one path per tree (true tree by walking the predecessors), at or below D
    def true_root(k):
        i=pred[k]; g=0
        while a[i]>=0:
            i=pred[a[i]]; g+=1
            assert g<=n
        return i
    best={}
    for k in np.where(freec&np.isfinite(d))[0]:
        r=true_root(k)
        if r not in best or d[k]<d[best[r]]: best[r]=k
    D=max(d[k] for k in best.values())
    inT=(d<D); p[inT]+=D-d[inT]
    for r,k in best.items():
        j=k
        while True:
            i=pred[j]; owner[j]=i; jp=a[i]; a[i]=j
            if i==r: break
            j=jp
    return len(best),batches,scans,dense

def solve(M,handoff=6,K=64,verbose=True):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    u=C.min(1); p=-(C-u[:,None]).min(0)
    a=np.full(n,-1); owner=np.full(n,-1); eps=Cr*8e-3; stats=[]
    while eps>=Cr*1e-6:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(0.02*n),stats); eps/=5
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,15,0,st)
    col_reduce(C,p,owner)
    F0=int((a<0).sum())
    cols,T=build_lists(C,p,min(K,n-1))
    ph=[]
    while (a<0).sum()>handoff:
        ph.append(forest_lists_phase(C,p,a,owner,np.where(a<0)[0],cols,T))
    free=np.where(a<0)[0]
    tb,ts,df,rc=sap_sparse(C,p,a,owner,free,cols,T)
    cost=C[np.arange(n),a].sum(); ref=C[lsa(C)].sum()
    ok=abs(cost-ref)<=1e-9*max(1,abs(ref)) and len(set(a))==n
    # dual feasibility certificate
    uu=C[np.arange(n),a]+p[a]; slack=(C+p[None,:]-uu[:,None]).min()
    if verbose: print(f"   n={n} free after ARR {F0}; forest phases (augmented,batches,scans,dense) {ph}; tail {len(free)} rows: {tb} batches, {df} dense; optimal {ok} minslack/Cr {slack/Cr:.1e}",flush=True)
    return ok, sum(x[1] for x in ph), tb

if __name__=="__main__":
    rng=np.random.default_rng(0)
    allok=True
    print("bench-like d=784"); 
    for kb in range(2):
        x0,x1=bench_batch(2048,784,1000,kb); ok,_,_=solve(cost32(x0,x1)); allok&=ok
    print("uniform random")
    for n in (300,1000): ok,_,_=solve(rng.random((n,n)).astype(np.float32)); allok&=ok
    print("geometric d=2 (deep paths)")
    for n in (256,700):
        x=rng.standard_normal((n,2)); y=rng.standard_normal((n,2))+0.5
        ok,_,_=solve(((x[:,None,:]-y[None])**2).sum(-1).astype(np.float32)); allok&=ok
    print("heavy ties (integer costs 0..9)")
    ok,_,_=solve(rng.integers(0,10,(400,400)).astype(np.float32)); allok&=ok
    print("ALL OPTIMAL" if allok else "MISMATCH")
