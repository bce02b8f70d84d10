import numpy as np, time, sys, itertools
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from proto9 import multi_phase
from scipy.optimize import linear_sum_assignment as lsa

def pipeline(C,Cr,cref,eps_last,arr_rounds,theta,stop,eps0=0.2,handoff=6):
    n=C.shape[0]
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*eps0; stats=[]
    while eps>=Cr*eps_last:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(stop*n),stats)
        eps/=theta
    R=sum(s[1] for s in stats); bids=sum(s[2] for s in stats)
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,arr_rounds,0,st)
    bids+=st[0][2]
    col_reduce(C,p,owner)
    F=np.sum(a<0)
    tr=0; per=[]; scans=0
    while np.sum(a<0)>handoff:
        k,rounds,sc,D=multi_phase(C,p,a,owner,"prune")
        tr+=rounds; per.append((k,rounds)); scans+=sc
    tail=int(np.sum(a<0))
    # finish to verify
    while np.any(a<0):
        multi_phase(C,p,a,owner,"prune")
    cost_rounds=R+st[0][1]+tr+13*tail
    print(f"el={eps_last:g} th={theta} stop={stop} arr={arr_rounds}: auction={R} free={F} dense_BF={tr} per={per} tail={tail} bids={bids} bfscans={scans} | COST={cost_rounds} mism={(a!=cref).sum()}",flush=True)
    return cost_rounds

if __name__=="__main__":
    n=int(sys.argv[1]); kb=int(sys.argv[2])
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    C=M.astype(np.float64); Cr=C.max()-C.min()
    r,cref=lsa(C)
    grid=[(el,arr,th,stop) for stop in [0.02,0.05,0.1,0.2] for th in [5.0,10.0,30.0] for el in [1e-6,1e-4] for arr in [5,15]]
    for (el,arr,th,stop) in grid:
        pipeline(C,Cr,cref,el,arr,th,stop)
