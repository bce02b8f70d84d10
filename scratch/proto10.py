import numpy as np, time, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from proto9 import multi_phase
from scipy.optimize import linear_sum_assignment as lsa

def pipeline(C,Cr,cref,eps_last,arr_rounds,theta=5.0,stop=0.02,eps0=0.2,tighten=False):
    n=C.shape[0]
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0; stats=[]
    while eps>=Cr*eps_last:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(stop*n),stats)
        col_reduce(C,p,owner)
        eps/=theta
    R=sum(s[1] for s in stats)
    st=[(0,0,0,0)]
    if arr_rounds>0:
        a[:]=-1; owner[:]=-1; st=[]
        auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    if tighten:
        u=(C+p[None,:]).min(1); asg=np.where(a>=0)[0]
        sl=C[asg,a[asg]]+p[a[asg]]-u[asg]
        bad=asg[sl>0]
        owner[a[bad]]=-1; a[bad]=-1
    col_reduce(C,p,owner)
    F=np.sum(a<0)
    tr=0; ts=0; ph=0; per=[]
    while np.any(a<0):
        k,rounds,scans,D=multi_phase(C,p,a,owner,"prune")
        tr+=rounds; ts+=scans; ph+=1; per.append((k,rounds))
    ar=np.arange(n); u=C[ar,a]+p[a]; S=C+p[None,:]-u[:,None]
    print(f"eps_last={eps_last:g} theta={theta} stop={stop} arr={arr_rounds} tighten={tighten}: auction rounds={R} arr={st[0][1]} free={F} | phases={ph} BF rounds={tr} scans={ts} per={per} | total rounds={R+st[0][1]+tr} mism={(a!=cref).sum()} minslack/Cr={S.min()/Cr:.1e}",flush=True)

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2])
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    C=M.astype(np.float64); Cr=C.max()-C.min()
    r,cref=lsa(C)
    pipeline(C,Cr,cref,1e-6,30)
    pipeline(C,Cr,cref,1e-6,10)
    pipeline(C,Cr,cref,1e-6,5)
    pipeline(C,Cr,cref,1e-6,0,tighten=True)
    pipeline(C,Cr,cref,1e-5,10)
    pipeline(C,Cr,cref,1e-4,10)
    pipeline(C,Cr,cref,1e-3,10)
    pipeline(C,Cr,cref,1e-7,10)
    pipeline(C,Cr,cref,1e-6,10,theta=10.0)
    pipeline(C,Cr,cref,1e-6,10,stop=0.05)
