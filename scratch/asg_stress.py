import sys, os, time, threading
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
rng=np.random.RandomState(0)
def instance(kind,n):
    if kind==0: return rng.rand(n,n).astype(np.float32)
    if kind==1:
        d=rng.choice([2,3,8,50,300]); a=rng.randn(n,d); b=rng.randn(n,d)*rng.uniform(0.3,2)+rng.uniform(-1,1)
        return ((a[:,None,:]-b[None,:,:])**2).sum(-1).astype(np.float32) if n*n*d<4e7 else (((a*a).sum(1)[:,None]+(b*b).sum(1)[None]-2*a@b.T).clip(0)).astype(np.float32)
    if kind==2: return rng.randint(0,20,(n,n)).astype(np.float32)           # heavy ties
    if kind==3: return (rng.rand(n,n)**8*1e4).astype(np.float32)            # skewed
    if kind==4:
        a=np.sort(rng.rand(n)); b=np.sort(rng.rand(n)); return ((a[:,None]-b[None])**2).astype(np.float32)  # 1-D: long chains
    if kind==5: return (-rng.rand(n,n)*1e3+rng.rand(n)[:,None]*1e3).astype(np.float32)
bad=0; tot=0; t0=time.time()
sizes=[2,3,37,64,100,255,256,513,1000,1500,2048]
def check(M):
    Mt=torch.from_numpy(M).to(dev)
    perm,info=ot.assign_exact(Mt,return_info=True)
    p=perm.cpu().numpy(); ref=oracle.exact_perm(M)
    c1,c2=oracle.assignment_cost(M,p),oracle.assignment_cost(M,ref)
    ok=(sorted(p.tolist())==list(range(len(p)))) and (c1==c2 or abs(c1-c2)<=1e-12*max(1,abs(c2)))
    return ok,(c1,c2,info['stats'])
for rep in range(4):
    for n in sizes:
        for kind in range(6):
            if kind==4 and n>1000: continue
            M=instance(kind,n)
            ok,inf=check(M); tot+=1
            if not ok: bad+=1; print("MISMATCH",kind,n,inf,flush=True)
print(f"sequential: {tot} instances, {bad} mismatches, {time.time()-t0:.1f}s",flush=True)
# concurrency: 3 threads on their own streams
errs=[]
def worker(seed):
    r=np.random.RandomState(seed); s=torch.cuda.Stream()
    with torch.cuda.stream(s):
        for k in range(12):
            n=int(r.choice([256,700,1024,2048])); M=r.rand(n,n).astype(np.float32) if k%2 else (r.randn(n,16)@r.randn(16,n)).astype(np.float32)
            ok,inf=check(M)
            if not ok: errs.append((seed,k,n,inf))
th=[threading.Thread(target=worker,args=(s,)) for s in (1,2,3)]
[t.start() for t in th]; [t.join() for t in th]
print("concurrent (3 threads x 12):", "OK" if not errs else errs)
