import numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import gen
from proto import auction_phase
from scipy.optimize import linear_sum_assignment as lsa

def build_lists(C,p,K):
    V=C+p[None,:]
    idx=np.argpartition(V,K,axis=1)[:,:K+1]          # K+1 smallest (unordered)
    vals=np.take_along_axis(V,idx,1)
    order=np.argsort(vals,1)
    idx=np.take_along_axis(idx,order,1); vals=np.take_along_axis(vals,order,1)
    T=vals[:,K].copy()                                # (K+1)-th smallest: all non-members >= T
    return idx[:,:K].copy(), T

def sap_sparse(C,p,a,owner,free_rows,cols,T,Kb=None):
    n=C.shape[0]; tot_batches=0; tot_scans=0; dense_fallbacks=0; dense_init=0; rechecks=0
    for i0 in free_rows:
        ck=cols[i0]; val=C[i0,ck]+p[ck]
        u0=val.min()
        dist=np.full(n,np.inf); pred=np.full(n,i0)
        dist[ck]=val-u0
        listed=np.zeros(n,bool); listed[ck]=owner[ck]>=0
        freec=owner<0
        dense_done=np.full(n,np.inf)   # per column j: base at which its row was last relaxed densely
        i0_dense=False
        batches=0
        while True:
            while True:
                dfree=dist[freec].min()
                S=np.where(listed&(dist<dfree))[0]
                if len(S)==0: break
                listed[S]=False
                newdist=dist.copy(); newpred=pred.copy()
                for j in S:
                    i=owner[j]; base=dist[j]; rj=C[i,j]+p[j]
                    ks=cols[i]; ks=ks[ks!=j]
                    cand=base+np.maximum((C[i,ks]+p[ks])-rj,0.0)
                    better=cand<newdist[ks]
                    kk=ks[better]; newdist[kk]=cand[better]; newpred[kk]=i
                imp=newdist<dist
                dist=newdist; pred=newpred
                listed|=imp&(owner>=0)
                batches+=1; tot_scans+=len(S)
            # posterior validity check
            rechecks+=1
            dfree=dist[freec].min()
            bad=[]
            if not i0_dense and not (T[i0]-u0>=dfree): bad.append(-1)
            tree=np.where((owner>=0)&(dist<dfree))[0]
            for j in tree:
                i=owner[j]; rj=C[i,j]+p[j]
                if dist[j]<dense_done[j] and not (dist[j]+(T[i]-rj)>=dfree): bad.append(j)
            if not bad: break
            # dense relax of the failing rows
            newdist=dist.copy(); newpred=pred.copy()
            for j in bad:
                dense_fallbacks+=1
                if j==-1:
                    i0_dense=True
                    cand=np.maximum((C[i0]+p)-u0,0.0); i=i0
                else:
                    i=owner[j]; base=dist[j]; rj=C[i,j]+p[j]; dense_done[j]=base
                    cand=base+np.maximum((C[i]+p)-rj,0.0); cand[j]=np.inf
                better=cand<newdist
                newdist[better]=cand[better]; newpred[better]=i
            imp=newdist<dist
            dist=newdist; pred=newpred
            listed|=imp&(owner>=0)
            batches+=1
        fc=np.where(freec)[0]; jf=fc[dist[fc].argmin()]; dfree=dist[jf]
        colsU=np.where((dist<dfree)&(owner>=0))[0]
        p[colsU]+=dfree-dist[colsU]
        j=jf
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j
            if i==i0: break
            j=jprev
        tot_batches+=batches
    return tot_batches,tot_scans,dense_fallbacks,rechecks

def run(M,cref,theta,eps0_frac,eps_last_frac,frac_stop,arr_rounds,K):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0_frac; stats=[]
    while eps>=Cr*eps_last_frac:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(frac_stop*n),stats)
        eps/=theta
    R=sum(s[1] for s in stats)
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    free=np.where(a<0)[0]
    cols,T=build_lists(C,p,K)
    V=C+p[None,:]; u=V.min(1)
    print("   margin T-u: min %.3e median %.3e ; eps_last %.3e"%((T-u).min(),np.median(T-u),Cr*eps_last_frac))
    tb,ts,df,di=sap_sparse(C,p,a,owner,free,cols,T)
    mism=(a!=cref).sum()
    u=(C+p[None,:])[np.arange(n),a]
    slack=C+p[None,:]-u[:,None]
    print(f"K={K} rounds={R} ARR={st[0][1]} free={len(free)} | batches={tb} rowscans={ts} dense_fallbacks={df} rechecks={di} | mismatch={mism} minslack={slack.min():.2e}")

if __name__=="__main__":
    cfg=sys.argv[1]; n=int(sys.argv[2])
    x0,x1=gen.get(cfg,n); M=gen.cost(x0,x1)
    t=time.time(); r,cref=lsa(M.astype(np.float64)); print("scipy",time.time()-t)
    for K in [64,32,16]:
        t=time.time(); run(M,cref,5,0.2,1e-6,0.02,30,K); print("   t=%.1fs"%(time.time()-t))
