import numpy as np, time, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
n=4096
for kb in [2,0]:
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1)
    C=M.astype(np.float64); Cr=C.max()-C.min()
    p=np.zeros(n); a=np.full(n,-1); owner=np.full(n,-1)
    eps=Cr*0.2; stats=[]
    while eps>=Cr*1e-6:
        a[:]=-1; owner[:]=-1
        auction_phase(C,p,a,owner,eps,100000,int(0.02*n),stats)
        col_reduce(C,p,owner)
        eps/=5
    print("batch",kb,"rounds",[s[1] for s in stats])
    # variant A: reset + eps=0 rounds, track free count
    pa=p.copy(); aa=np.full(n,-1); oa=np.full(n,-1)
    tot=0
    for cap in [10,20,30,60,100,200,400,800]:
        st=[]; auction_phase(C,pa,aa,oa,0.0,cap-tot,0,st); tot=cap
        print(f"   reset ARR rounds={cap}: free={np.sum(aa<0)}")
    # variant B: keep the assignment of the last phase (eps-CS), eps=0 rounds only for free rows... (no reset)
    pb=p.copy(); ab=a.copy(); ob=owner.copy(); tot=0
    for cap in [10,20,30,60,100,200,400,800]:
        st=[]; auction_phase(C,pb,ab,ob,0.0,cap-tot,0,st); tot=cap
        u=(C+pb[None,:]).min(1); asg=np.where(ab>=0)[0]
        sl=C[asg,ab[asg]]+pb[ab[asg]]-u[asg]
        print(f"   noreset eps=0 rounds={cap}: free={np.sum(ab<0)} nontight assigned={np.sum(sl>1e-12*Cr)}")
