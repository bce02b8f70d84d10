# LAPMOD-style: sparse SAP on lists with NO per-search validity test; dense pricing pass at the end.
import numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from proto import auction_phase
from proto4 import build_lists
from proto5 import bench_batch, cost32
from scipy.optimize import linear_sum_assignment as lsa

def sap_lists(C,p,a,owner,free_rows,cols):
    """label-correcting SAP restricted to the list edges; returns batches, scans, dfree list"""
    n=C.shape[0]; tb=0; ts=0; dfs=[]; sizes=[]
    for i0 in free_rows:
        ck=cols[i0]; val=C[i0,ck]+p[ck]; u0=val.min()
        dist=np.full(n,np.inf); pred=np.full(n,i0)
        dist[ck]=val-u0
        listed=np.zeros(n,bool); listed[ck]=owner[ck]>=0
        freec=owner<0
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
            tb+=1; ts+=len(S)
        fc=np.where(freec)[0]; jf=fc[dist[fc].argmin()]; dfree=dist[jf]
        if not np.isfinite(dfree): raise RuntimeError("no path on lists")
        colsU=np.where((dist<dfree)&(owner>=0))[0]
        sizes.append(len(colsU)); dfs.append(dfree)
        p[colsU]+=dfree-dist[colsU]
        j=jf
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j
            if i==i0: break
            j=jprev
    return tb,ts,dfs,sizes

def run(M,cref,K,theta=5,eps0_frac=0.2,eps_last_frac=1e-6,frac_stop=0.02,arr_rounds=30):
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
    u=(C+p[None,:]).min(1)
    print(f"  rounds={R} free={len(free)} margin T-u: min {(T-u).min()/Cr:.3e} med {np.median(T-u)/Cr:.3e} (xCr)")
    it=0
    while True:
        tb,ts,dfs,sizes=sap_lists(C,p,a,owner,free,cols)
        print(f"  iter{it}: searches={len(free)} batches={tb} scans={ts} sum_dfree/Cr={sum(dfs)/Cr:.3e} max_dfree/Cr={max(dfs)/Cr:.3e} tree sizes max={max(sizes)} mean={np.mean(sizes):.0f}")
        # dense pricing
        ar=np.arange(n)
        u=C[ar,a]+p[a]
        S=C+p[None,:]-u[:,None]
        viol=S< -1e-12*Cr
        vr=np.where(viol.any(1))[0]
        print(f"     pricing: violating rows={len(vr)} edges={viol.sum()} minslack/Cr={S.min()/Cr:.3e} mism={(a!=cref).sum()}")
        if len(vr)==0: break
        # add violated edges to lists (append), unassign those rows
        newcols=[]
        for i in range(n): newcols.append(cols[i])
        for i in vr:
            ks=np.where(viol[i])[0]
            newcols[i]=np.unique(np.concatenate([cols[i],ks]))
            j=a[i]; owner[j]=-1; a[i]=-1
        cols=newcols
        free=vr; it+=1
        if it>10: break

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2]); K=int(sys.argv[3]) if len(sys.argv)>3 else 64
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    t=time.time(); r,cref=lsa(M.astype(np.float64)); print("scipy",time.time()-t)
    for K in [64,32]:
        t=time.time(); run(M,cref,K); print("   t=%.1fs"%(time.time()-t))
