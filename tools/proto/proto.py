import numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
import gen
from scipy.optimize import linear_sum_assignment as lsa

def top2(R):
    # R: [m,n] -> best idx, best val, second val
    j1 = R.argmin(1); ar=np.arange(R.shape[0])
    b = R[ar,j1]
    R2 = R.copy(); R2[ar,j1]=np.inf
    s = R2.min(1)
    return j1,b,s

def auction_phase(C,p,a,owner,eps,max_rounds,stop_unassigned,stats):
    n=C.shape[0]
    U=np.where(a<0)[0]
    rounds=0; bids=0
    while len(U)>stop_unassigned and rounds<max_rounds:
        R=C[U]+p[None,:]
        j1,b,s=top2(R)
        bid=p[j1]+(s-b)+eps
        # winner per object: max bid
        order=np.lexsort((U,-bid))  # sort by bid desc
        jj=j1[order]; first=np.unique(jj,return_index=True)[1]
        w=order[first]              # indices into U of winners
        wi=U[w]; wj=j1[w]
        prev=owner[wj]
        p[wj]=bid[w]
        a[prev[prev>=0]]=-1
        owner[wj]=wi; a[wi]=wj
        bids+=len(U); rounds+=1
        U=np.where(a<0)[0]
    stats.append((eps,rounds,bids,len(U)))
    return rounds,bids

def sap(C,u,v,a,owner,free_rows):
    # JV augmentation in fp64. returns total steps
    n=C.shape[0]; steps=[]
    for i0 in free_rows:
        dist=C[i0]-u[i0]-v
        pred=np.full(n,i0)
        scanned=np.zeros(n,bool)
        k=0
        while True:
            dm=np.where(scanned,np.inf,dist)
            j=int(dm.argmin()); dj=dist[j]
            scanned[j]=True; k+=1
            if owner[j]<0: break
            i=owner[j]
            nd=dj+(C[i]-u[i]-v) - (C[i,j]-u[i]-v[j])
            upd=(~scanned)&(nd<dist)
            dist[upd]=nd[upd]; pred[upd]=i
        steps.append(k)
        # dual update
        sc=scanned.copy(); sc[j]=False
        # columns scanned before final (dist<dj)
        cols=np.where(sc)[0]
        u[i0]+=dj
        for c in cols:
            u[owner[c]]+=dj-dist[c]
        v[cols]-=dj-dist[cols]
        # augment
        while True:
            i=pred[j]; owner[j]=i; jprev=a[i]; a[i]=j
            if i==i0: break
            j=jprev
    return steps

def solve(C32,theta=5.0,eps0_frac=0.2,eps_final_frac=1e-6,arr_rounds=50,frac_stop=0.02,verbose=True):
    C=C32.astype(np.float64); n=C.shape[0]
    Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0_frac; stats=[]
    tot_rounds=0; tot_bids=0
    while True:
        last = eps<=Cr*eps_final_frac
        a[:]=-1; owner[:]=-1
        r,b=auction_phase(C,p,a,owner,eps,100000, 0 if last else int(frac_stop*n),stats)
        tot_rounds+=r; tot_bids+=b
        if last: break
        eps=max(eps/theta,Cr*eps_final_frac)
    if verbose:
        for s in stats: print("  eps=%.3e rounds=%d bids=%d left=%d"%s)
    # check optimality of auction assignment
    return C,p,a,owner,tot_rounds,tot_bids

if __name__=="__main__":
    cfg=sys.argv[1]; n=int(sys.argv[2])
    x0,x1=gen.get(cfg,n); M=gen.cost(x0,x1)
    t=time.time(); r,cref=lsa(M.astype(np.float64)); print("scipy",time.time()-t)
    for ef in [1e-4,1e-6,1e-8]:
        t=time.time()
        C,p,a,owner,R,Bd=solve(M,eps_final_frac=ef,verbose=(ef==1e-6))
        print("eps_final_frac",ef,"rounds",R,"bids",Bd,"bids/n",Bd/n,"mismatch vs scipy",(a!=cref).sum(),"time",time.time()-t)
        # ARR eps=0 from scratch with these prices
        a2=np.full(n,-1); own2=np.full(n,-1); st=[]; p2=p.copy()
        auction_phase(C,p2,a2,own2,0.0,30,0,st)
        print("   ARR:",st)
        free=np.where(a2<0)[0]
        v=-p2; u=(C+p2[None,:]).min(1)
        # verify tightness of assigned
        asg=np.where(a2>=0)[0]
        sl=C[asg,a2[asg]]-u[asg]-v[a2[asg]]
        print("   max slack assigned",sl.max() if len(sl) else 0,"free",len(free))
        steps=sap(C,u,v,a2,own2,free)
        print("   SAP steps total",sum(steps),"max",max(steps) if steps else 0,"mismatch",(a2!=cref).sum())
