import numpy as np, sys
sys.path.insert(0,'scratch')
from proto import auction_phase
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from proto9 import multi_phase
n=4096
for kb in [2,5]:
    x0,x1=bench_batch(n,784,1000,kb); M=cost32(x0,x1); C=M.astype(np.float64); Cr=C.max()-C.min()
    u=C.min(1); p0=-(C-u[:,None]).min(0)
    def run(stops,tag):
        p=p0.copy(); a=np.full(n,-1); owner=np.full(n,-1); eps=Cr*8e-3; stats=[]; ph=0
        while eps>=Cr*1e-6:
            a[:]=-1; owner[:]=-1
            auction_phase(C,p,a,owner,eps,100000,int(stops[min(ph,len(stops)-1)]*n),stats)
            eps/=5; ph+=1
        a[:]=-1; owner[:]=-1; st=[]
        auction_phase(C,p,a,owner,0.0,15,0,st)
        col_reduce(C,p,owner)
        R=sum(s[1] for s in stats); F=int(np.sum(a<0))
        tr=0; per=[]
        while np.sum(a<0)>6:
            k,rounds,sc,D=multi_phase(C,p,a,owner,"full"); tr+=rounds; per.append((k,rounds))
        tail=int(np.sum(a<0))
        print(f"batch{kb} {tag}: rounds {[s[1] for s in stats]} total {R} free {F} MS {per} tail {tail}  score {R+tr+13*tail}",flush=True)
    run([0.02],"stop .02")
    run([0.05,0.05,0.04,0.02,0.02,0.02],"early .05")
    run([0.08,0.06,0.04,0.02,0.02,0.02],"early .08")
    run([0.05,0.05,0.05,0.05,0.02,0.02],"4x .05")
    run([0.05,0.05,0.05,0.05,0.05,0.02],"5x .05")
