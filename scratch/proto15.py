import numpy as np, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
n=4096
for kb in [2,5]:
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1); C=M.astype(np.float64); Cr=C.max()-C.min()
    u=C.min(1); p0=-(C-u[:,None]).min(0)
    def run(eps0,theta,elast,stop=0.02,arr=15):
        p=p0.copy(); a=np.full(n,-1); owner=np.full(n,-1); eps=Cr*eps0; stats=[]
        while eps>=Cr*elast:
            a[:]=-1; owner[:]=-1
            auction_phase(C,p,a,owner,eps,100000,int(stop*n),stats)
            eps/=theta
        a[:]=-1; owner[:]=-1; st=[]
        auction_phase(C,p,a,owner,0.0,arr,0,st)
        R=sum(s[1] for s in stats)
        print(f"batch{kb} eps0={eps0:g} theta={theta} elast={elast:g} stop={stop}: rounds {[s[1] for s in stats]} total {R} free {st[0][3]}  score {R+2.5*st[0][3]:.0f}",flush=True)
    run(8e-3,5,1e-6)
    run(8e-3,8,1e-6)
    run(8e-3,12,1e-6)
    run(4e-3,6,1e-6)
    run(3e-3,10,1e-6)
    run(8e-3,5,1e-5)
    run(8e-3,8,1e-5)
    run(2e-3,5,1e-6)
    run(8e-3,5,1e-6,stop=0.03)
