import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
lib=_lib.load(); dev=_lib.require_gpu()
B,d=4096,784
Ms=[]
for seed in (1000,2000,3000,4000,5000):
    for (x0,x1) in bench.synth_batches(B,d,8,seed,dev):
        Ms.append(ot.cost_matrix(x0,x1,matrix_cores=False))
def run(tag):
    ms=[]; sap=[]
    for M in Ms:
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); perm,info=ot.assign_exact(M,return_info=True); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1)); sap.append(info["stats"][3])
    a=np.array(ms)
    print(f"{tag:34s} all40 {a.mean():.3f} +- {a.std()/np.sqrt(len(a)):.3f} ms | bench pool (first 8) {a[:8].mean():.3f} | sapb {np.mean(sap):.0f} | max {a.max():.2f}",flush=True)
def P(theta=5.0,eps0=8e-3,epsl=1e-6,stop=0.02,rc=4000,arr=15): lib.cfm_assign_set_params(theta,eps0,epsl,stop,rc,arr,64)
run("warmup"); run("default (handoff 6)")
for h in (0,2,4,10,16,32):
    lib.cfm_assign_set_handoff(h); run(f"handoff {h}")
lib.cfm_assign_set_handoff(6)
lib.cfm_assign_set_mode(0); run("dense (no sparse solver)"); lib.cfm_assign_set_mode(1)
for arr in (5,10,25,40): P(arr=arr); run(f"arr_cap {arr}")
P()
for stop in (0.01,0.04): P(stop=stop); run(f"stop_frac {stop}")
P()
for q in (0.5,0.75): lib.cfm_assign_set_ms_quantile(q); run(f"ms_q {q}")
lib.cfm_assign_set_ms_quantile(1.0)
for h,arr in ((16,25),(32,25),(16,10)):
    lib.cfm_assign_set_handoff(h); P(arr=arr); run(f"handoff {h} arr {arr}")
