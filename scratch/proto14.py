import numpy as np, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
n=4096
x0,x1=bench_batch(n,784,1000,2); M=cost32(x0,x1); C=M.astype(np.float64); Cr=C.max()-C.min()
def run(p0,eps0,tag,theta=5.0):
    p=p0.copy(); a=np.full(n,-1); owner=np.full(n,-1); eps=Cr*eps0; stats=[]
    while eps>=Cr*1e-6:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(0.02*n),stats)
        eps/=theta
    a[:]=-1; owner[:]=-1; st=[]
    auction_phase(C,p,a,owner,0.0,15,0,st)
    print(tag,"rounds",[s[1] for s in stats],"total",sum(s[1] for s in stats),"bids",sum(s[2] for s in stats),"free after ARR",st[0][3])
run(np.zeros(n),0.2,"zero prices eps0=0.2")
pc=-C.min(0)
for e0 in [0.2,0.04,8e-3,1.6e-3]:
    run(pc,e0,f"colmin prices eps0={e0}")
# row-then-col reduction style: u_i=min_j c_ij ; p_j = -min_i (c_ij - u_i)
u=C.min(1); pc2=-(C-u[:,None]).min(0)
for e0 in [0.04,8e-3,1.6e-3]:
    run(pc2,e0,f"row+col reduced prices eps0={e0}")
