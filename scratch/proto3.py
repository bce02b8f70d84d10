import numpy as np, time, sys
sys.path.insert(0,'.')
import gen
from proto import auction_phase
from scipy.optimize import linear_sum_assignment as lsa

def sap_bf(C,u,v,a,owner,free_rows,K=None):
    n=C.shape[0]; tot_batches=0; tot_scans=0; per=[]
    for i0 in free_rows:
        dist=np.maximum(C[i0]-u[i0]-v,0.0)
        pred=np.full(n,i0)
        dirty=owner>=0
        freec=owner<0
        batches=0; scans=0
        while True:
            dfree=dist[freec].min()
            S=np.where(dirty&(dist<dfree))[0]
            if len(S)==0: break
            if K is not None and len(S)>K:
                S=S[np.argsort(dist[S])[:K]]
            dirty[S]=False
            rows=owner[S]
            RC=np.maximum(C[rows]-u[rows][:,None]-v[None,:],0.0)   # [|S|,n]
            cand=dist[S][:,None]+RC
            cand[np.arange(len(S)),S]=np.inf
            m=cand.argmin(0); cm=cand[m,np.arange(n)]
            upd=cm<dist
            dist[upd]=cm[upd]; pred[upd]=rows[m[upd]]; dirty[upd&(owner>=0)]=True
            batches+=1; scans+=len(S)
        jf=np.where(freec)[0][dist[freec].argmin()]; dfree=dist[jf]
        cols=np.where((dist<dfree)&(owner>=0))[0]
        u[i0]+=dfree
        u[owner[cols]]+=dfree-dist[cols]
        v[cols]-=dfree-dist[cols]
        j=jf; hops=0
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j; hops+=1
            if i==i0: break
            j=jprev
        tot_batches+=batches; tot_scans+=scans; per.append((batches,scans,hops))
    return tot_batches,tot_scans,per

def run(M,cref,theta,eps0_frac,eps_last_frac,frac_stop,arr_rounds,K):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0_frac; stats=[]
    while eps>=Cr*eps_last_frac:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(frac_stop*n),stats)
        eps/=theta
    R=sum(s[1] for s in stats); Bd=sum(s[2] for s in stats)
    full=sum(1 for s in stats)  # first round of each phase is full
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    free=np.where(a<0)[0]
    v=-p; u=(C+p[None,:]).min(1)
    tb,ts,per=sap_bf(C,u,v,a,owner,free,K)
    mism=(a!=cref).sum()
    # certificate
    slack=C-u[:,None]-v[None,:]
    print(f"theta={theta} elast={eps_last_frac:g} stop={frac_stop} K={K} | phases={len(stats)} rounds={R} bids/n={Bd/n:.1f} | ARR rounds={st[0][1]} free={len(free)} | BF batches={tb} rowscans={ts} maxhops={max(p_[2] for p_ in per) if per else 0} | mismatch={mism} minslack={slack.min():.2e} maxslack_assigned={np.abs(slack[np.arange(n),a]).max():.2e}")
    return per

if __name__=="__main__":
    cfg=sys.argv[1]; n=int(sys.argv[2])
    x0,x1=gen.get(cfg,n); M=gen.cost(x0,x1)
    t=time.time(); r,cref=lsa(M.astype(np.float64)); print("scipy",time.time()-t)
    for (theta,el,fs,ar,K) in [(5,1e-6,0.02,30,None),(5,1e-6,0.02,30,128),(5,1e-6,0.02,30,32),(5,1e-8,0.02,30,None),(5,1e-4,0.02,30,None)]:
        t=time.time(); per=run(M,cref,theta,0.2,el,fs,ar,K); print("   t=%.1fs"%(time.time()-t), per[:6])
