import sys, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
acc = np.zeros(12); n = 0
for seed in range(8):
    a, b = oracle.config_inputs("C1", rank=seed)
    M = ot.cost_matrix(a.to(dev), b.to(dev), matrix_cores=False)
    for rep in range(6):
        ot.assign_exact(M); torch.cuda.synchronize()
        buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); acc += np.array(list(buf)[4:16]) / 100.0; n += 1
print("phase us:", " ".join(f"{v:.0f}" for v in acc / n), "sum", f"{acc.sum()/n:.0f}")
