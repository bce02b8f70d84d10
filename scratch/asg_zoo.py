import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
def run(name,M,check=True):
    for r in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter()
        perm,info=ot.assign_exact(M,return_info=True)
        torch.cuda.synchronize(); dt=1e3*(time.perf_counter()-t0)
    s=info['stats']; ok=''
    if check:
        Mn=M.cpu().numpy(); p=perm.cpu().numpy(); ref=oracle.exact_perm(Mn)
        ok=f"same_perm={np.array_equal(p,ref)} cost_equal={oracle.assignment_cost(Mn,p)==oracle.assignment_cost(Mn,ref)}"
    print(f"{name}: {dt:.2f} ms rounds={s[0]} arr={s[1]} free={s[2]} sap_batches={s[3]} steps={s[6]} ms_phases={(s[7]>>8)&255} fallbacks={s[7]>>16} {ok}",flush=True)
g=torch.Generator().manual_seed(0)
x0,x1=oracle.config_inputs("C2"); run("C2 d=2 B=4096",ot.cost_matrix(x0.to(dev),x1.to(dev)))
run("uniform random 4096",torch.rand(4096,4096,generator=g).to(dev))
run("integer costs 0..9 (heavy ties) 2048",torch.randint(0,10,(2048,2048),generator=g).float().to(dev),check=True)
x=torch.randn(4096,50,generator=g); y=torch.randn(4096,50,generator=g)*1.5+0.5
run("gauss d=50 B=4096",ot.cost_matrix(x.to(dev),y.to(dev)))
x=torch.randn(1000,3,generator=g); run("identical sets (zero diagonal) 1000",ot.cost_matrix(x.to(dev),x.to(dev)))
x0,x1=oracle.config_inputs("C3",B=8192); run("C3 d=784 B=8192",ot.cost_matrix(x0.to(dev),x1.to(dev)),check=False)
x0,x1=oracle.config_inputs("C3",B=1000); run("C3 d=784 B=1000",ot.cost_matrix(x0.to(dev),x1.to(dev)))
