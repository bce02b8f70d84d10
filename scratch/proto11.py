import numpy as np, time, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from proto9 import multi_phase
from scipy.optimize import linear_sum_assignment as lsa

def pipeline(C,Cr,cref,colred_from,arr_rounds=30,eps_last=1e-6,theta=5.0,stop=0.02,eps0=0.2,switch_at=6):
    n=C.shape[0]
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0; stats=[]; ph=0
    while eps>=Cr*eps_last:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(stop*n),stats)
        if ph>=colred_from: col_reduce(C,p,owner)
        eps/=theta; ph+=1
    R=sum(s[1] for s in stats)
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    col_reduce(C,p,owner)
    F=np.sum(a<0)
    tr=0; per=[]
    while np.any(a<0):
        k,rounds,scans,D=multi_phase(C,p,a,owner,"prune")
        tr+=rounds; per.append((k,rounds))
    print(f"colred_from={colred_from} arr={arr_rounds}: auction rounds={R} {[s[1] for s in stats]} free={F} | BF rounds={tr} per={per} mism={(a!=cref).sum()}",flush=True)

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2])
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    C=M.astype(np.float64); Cr=C.max()-C.min()
    r,cref=lsa(C)
    for cf in [99,7,6,4,0]:
        pipeline(C,Cr,cref,cf,30)
    pipeline(C,Cr,cref,99,15); pipeline(C,Cr,cref,6,15)
