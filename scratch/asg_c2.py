import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib=_lib.load(); dev=_lib.require_gpu()
x0,x1=oracle.config_inputs("C2"); M=ot.cost_matrix(x0.to(dev),x1.to(dev))
def run(tag):
    for r in range(2):
        torch.cuda.synchronize(); t0=time.perf_counter()
        perm,info=ot.assign_exact(M,return_info=True)
        torch.cuda.synchronize(); dt=1e3*(time.perf_counter()-t0)
    s=info['stats']
    print(f"{tag}: {dt:.2f} ms rounds={s[0]} arr={s[1]} free={s[2]} sap_batches={s[3]} steps={s[6]} ms_phases={(s[7]>>8)&255} fallbacks={s[7]>>16}",flush=True)
# set_params(theta, eps0_frac, eps_last_frac, stop_frac, round_cap, arr_cap, chunk)
run("default")
lib.cfm_assign_set_params(0.0,0.0,0.0,-1.0,40,15,0); run("round_cap 40")
lib.cfm_assign_set_params(0.0,0.0,0.0,-1.0,20,15,0); run("round_cap 20")
lib.cfm_assign_set_params(0.0,0.0,0.0,-1.0,40,15,0); lib.cfm_assign_set_handoff(64); run("round_cap 40 handoff 64")
lib.cfm_assign_set_handoff(0); run("round_cap 40 handoff 0 (dense only)")
lib.cfm_assign_set_handoff(6); lib.cfm_assign_set_ms_quantile(0.5); run("round_cap 40 q 0.5")
lib.cfm_assign_set_ms_quantile(0.25); run("round_cap 40 q 0.25")
lib.cfm_assign_set_ms_quantile(1.0); lib.cfm_assign_set_params(0.0,0.0,0.0,0.05,40,15,0); run("round_cap 40 stop .05")
lib.cfm_assign_set_params(10.0,0.0,0.0,0.02,40,15,0); run("round_cap 40 theta 10")
