# effect of a reverse "column reduction" of the free columns before the SAP phase
import numpy as np, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from proto import auction_phase
from proto4 import build_lists
from proto5 import bench_batch, cost32
from proto6 import sap_lists
from scipy.optimize import linear_sum_assignment as lsa

def col_reduce(C,p,owner):
    u=(C+p[None,:]).min(1)
    fc=np.where(owner<0)[0]
    newp=(u[:,None]-C[:,fc]).max(0)
    dec=p[fc]-newp
    p[fc]=newp
    return dec

def run(M,cref,K,colred,phase_colred=False,theta=5,eps0_frac=0.2,eps_last_frac=1e-6,frac_stop=0.02,arr_rounds=30):
    C=M.astype(np.float64); n=C.shape[0]; Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0_frac; stats=[]
    while eps>=Cr*eps_last_frac:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(frac_stop*n),stats)
        if phase_colred: col_reduce(C,p,owner)
        eps/=theta
    R=sum(s[1] for s in stats)
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    free=np.where(a<0)[0]
    if colred:
        dec=col_reduce(C,p,owner)
        print(f"  col reduction: price drops/Cr mean {dec.mean()/Cr:.3e} max {dec.max()/Cr:.3e}")
    cols,T=build_lists(C,p,K)
    u=(C+p[None,:]).min(1)
    print(f"  rounds={R} {[s[1] for s in stats]} free={len(free)} margin T-u: min {(T-u).min()/Cr:.3e} med {np.median(T-u)/Cr:.3e} (xCr)")
    it=0
    while True:
        tb,ts,dfs,sizes=sap_lists(C,p,a,owner,free,cols)
        print(f"  iter{it}: searches={len(free)} batches={tb} scans={ts} sum_dfree/Cr={sum(dfs)/Cr:.3e} max_dfree/Cr={max(dfs)/Cr:.3e} tree sizes max={max(sizes)} mean={np.mean(sizes):.0f}")
        ar=np.arange(n)
        u=C[ar,a]+p[a]
        S=C+p[None,:]-u[:,None]
        viol=S< -1e-12*Cr
        vr=np.where(viol.any(1))[0]
        print(f"     pricing: violating rows={len(vr)} edges={viol.sum()} minslack/Cr={S.min()/Cr:.3e} mism={(a!=cref).sum()}")
        if len(vr)==0: break
        newcols=[cols[i] for i in range(n)]
        for i in vr:
            ks=np.where(viol[i])[0]
            newcols[i]=np.unique(np.concatenate([cols[i],ks]))
            j=a[i]; owner[j]=-1; a[i]=-1
        cols=newcols
        free=vr; it+=1
        if it>10: break

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2])
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    t=time.time(); r,cref=lsa(M.astype(np.float64)); print("scipy",time.time()-t)
    for (cr,pcr) in [(False,False),(True,False),(True,True)]:
        print("colred",cr,"phase_colred",pcr)
        t=time.time(); run(M,cref,64,cr,pcr); print("   t=%.1fs"%(time.time()-t))
