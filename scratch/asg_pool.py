import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
lib=_lib.load(); dev=_lib.require_gpu()
nb=int(sys.argv[1]) if len(sys.argv)>1 else 8
if len(sys.argv)>2: lib.cfm_assign_set_handoff(int(sys.argv[2]))
stop=float(sys.argv[4]) if len(sys.argv)>4 else -1.0
theta=float(sys.argv[5]) if len(sys.argv)>5 else 0.0
elast=float(sys.argv[6]) if len(sys.argv)>6 else 0.0
if len(sys.argv)>3: lib.cfm_assign_set_params(theta,0.0,elast,stop,0,int(sys.argv[3]),0)
import cfm_oracle as oracle
if os.environ.get('STOPE'): lib.cfm_assign_set_stop_early(float(os.environ['STOPE']))
if os.environ.get('MSQ'): lib.cfm_assign_set_ms_quantile(float(os.environ['MSQ']))
pool=bench.synth_batches(4096,784,nb,1000,dev)
side=torch.cuda.Stream() if os.environ.get('SIDE') else torch.cuda.current_stream()
torch.cuda.synchronize()
ctx=torch.cuda.stream(side)
ctx.__enter__()
for k,(x0,x1) in enumerate(pool):
    M=ot.cost_matrix(x0,x1)
    for r in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter()
        perm,info=ot.assign_exact(M,return_info=True)
        torch.cuda.synchronize(); dt=1e3*(time.perf_counter()-t0)
    s=info['stats']
    print(f"batch {k}: {dt:.2f} ms auction_rounds={s[0]} arr={s[1]} free={s[2]} sap_batches={s[3]} sap_scans={s[4]} total_scans={s[5]} steps={s[6]} phase={s[7]&255} ms_phases={(s[7]>>8)&255} dense_fallbacks={s[7]>>16}",flush=True)
    if k==0 and os.environ.get('CHECK'):
        ok=np.array_equal(perm.cpu().numpy(), oracle.exact_perm(M.cpu().numpy())); print("   parity vs scipy:",ok,flush=True)
