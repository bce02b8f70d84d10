import sys, os, ctypes
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'oracle'))
import torch, numpy as np
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
if os.environ.get('CFM_LIB_OVERRIDE'): _lib.LIB_PATH=os.environ['CFM_LIB_OVERRIDE']
lib=_lib.load(); dev=_lib.require_gpu()
B,d=4096,784
names=["INIT","AUCTION","ARR","SAP(ms relax)","CERT","DONE","BUILD","SAP1","SAP1_DONE(solver)","UMIN","COLRED","ROOTMIN","UMIN0","INITRED"]
Ms=[]
for seed in (1000,2000,3000,4000,5000):
    for (x0,x1) in bench.synth_batches(B,d,8,seed,dev):
        Ms.append(ot.cost_matrix(x0,x1,matrix_cores=False))
ws=_lib.workspace(_lib.OP_ASSIGN,B,B,0,dev)
acc=np.zeros(32); tot=[]; evs=[]; chk=0
for rep in range(2):
    for M in Ms:
        torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); perm,info=ot.assign_exact(M,return_info=True); e1.record(); torch.cuda.synchronize()
        if rep==1:
            chk=(chk*1000003+int((perm.long()*torch.arange(1,B+1,device=dev)).sum().item()))%(2**61-1)
            buf=(ctypes.c_double*32)(); _lib.check(lib.cfm_assign_debug_times(_lib.ptr(ws),buf),"dbg")
            t=np.array(list(buf)); acc+=t; tot.append(t[:16].sum()); evs.append(e0.elapsed_time(e1)*1e3)
n=len(Ms)
print("perm checksum",chk); print(f"mean solve (events) {np.mean(evs):.0f} us; booked by the controller {np.mean(tot):.0f} us")
for q,nm in enumerate(names):
    if acc[q]>0: print(f"  {nm:20s} {acc[q]/n:8.1f} us  {100*acc[q]/acc[:16].sum():5.1f} %   of which controller {acc[16+q]/n:7.1f} us")
