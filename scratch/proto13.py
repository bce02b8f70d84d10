# radius rule study: D_est = q-quantile of the current free-column labels (q=1: max)
import numpy as np, time, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from scipy.optimize import linear_sum_assignment as lsa

def multi_phase_q(C,p,a,owner,q):
    n=C.shape[0]
    free_rows=np.where(a<0)[0]; freec=owner<0; fcs=np.where(freec)[0]
    V=C[free_rows]+p[None,:]
    u0=V.min(1)
    d=np.full(n,np.inf); pred=np.full(n,-1)
    # list entries: (row, col or -1, base, rj)
    rows=free_rows.copy(); cols=np.full(len(rows),-1); base=np.zeros(len(rows)); rj=u0.copy()
    rounds=0; scans=0; hist=[]
    while True:
        lab=np.sort(d[fcs])
        kq=max(0,int(np.ceil(q*len(fcs)))-1)
        D=lab[kq] if len(rows)>0 else np.inf
        if len(fcs)==1 or len(free_rows)==1: D=lab[0]
        act=base<D
        if not act.any(): break
        r_=rows[act]; c_=cols[act]; b_=base[act]; j_=rj[act]
        RC=np.maximum(C[r_]+p[None,:]-j_[:,None],0.0)
        cand=b_[:,None]+RC
        m=cols[act]>=0
        cand[np.where(m)[0],c_[m]]=np.inf
        am=cand.argmin(0); cm=cand[am,np.arange(n)]
        upd=cm<d
        d[upd]=cm[upd]; pred[upd]=r_[am[upd]]
        nxt=np.where(upd&(owner>=0)&(d<D))[0]      # appended only below the radius in force
        rounds+=1; scans+=int(act.sum()); hist.append(int(act.sum()))
        rows=owner[nxt]; cols=nxt; base=d[nxt]; rj=C[rows,nxt]+p[nxt]
    # accept per tree
    lab=d[fcs]; Dq=np.sort(lab)[max(0,int(np.ceil(q*len(fcs)))-1)]
    if len(fcs)==1 or len(free_rows)==1: Dq=lab.min()
    best={}
    for k in fcs:
        if not (d[k]<=Dq): continue
        i=pred[k]; g=0
        while a[i]>=0:
            i=pred[a[i]]; g+=1
        if i not in best or d[k]<d[best[i]]: best[i]=k
    D=max(d[k] for k in best.values())
    inT=(d<D); p[inT]+=D-d[inT]
    for r,k in best.items():
        j=k
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j
            if i==r: break
            j=jprev
    return len(best),rounds,scans,hist

def prep(C,Cr,stop=0.02,arr=15):
    n=C.shape[0]
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*0.2; stats=[]
    while eps>=Cr*1e-6:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(stop*n),stats)
        eps/=5
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr,0,st)
    col_reduce(C,p,owner)
    return p,a,owner

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2]); stop=float(sys.argv[3]) if len(sys.argv)>3 else 0.02
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    C=M.astype(np.float64); Cr=C.max()-C.min()
    r,cref=lsa(C)
    p0,a0,o0=prep(C,Cr,stop)
    for q in [1.0,0.75,0.5,0.25,0.1]:
        p=p0.copy(); a=a0.copy(); owner=o0.copy()
        tr=0; ts=0; per=[]
        while np.sum(a<0)>6:
            k,rounds,scans,hist=multi_phase_q(C,p,a,owner,q)
            tr+=rounds; ts+=scans; per.append((k,rounds,scans,hist[:6]))
        tail=int(np.sum(a<0))
        while np.any(a<0): multi_phase_q(C,p,a,owner,1.0)
        ar=np.arange(n); u=C[ar,a]+p[a]; S=C+p[None,:]-u[:,None]
        print(f"q={q}: free0={np.sum(a0<0)} phases={len(per)} rounds={tr} scans={ts} tail={tail} mism={(a!=cref).sum()} minslack={S.min()/Cr:.1e}\n     {per}")
