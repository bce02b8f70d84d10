"""Phase C statistics of the one-workgroup solver over C1-like instances (free rows, searches, scans, steps, phase times)."""
import sys, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
Ms = []
for seed in range(8):
    a, b = oracle.config_inputs("C1", rank=seed)
    Ms.append(ot.cost_matrix(a.to(dev), b.to(dev), matrix_cores=False))
acc = np.zeros(10); cnt = 0
for rep in range(12):
    for M in Ms:
        ot.assign_exact(M); torch.cuda.synchronize()
        buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); b = list(buf)
        acc += np.array([b[2], b[13], b[14], 0, 0, b[15] / 100, (b[5] - b[15]) / 100, b[7] / 100, b[9], (b[4] + b[5] + b[6] + b[7] + b[8]) / 100]); cnt += 1
m = acc / cnt
print(f"free {m[0]:.2f} searches {m[1]:.2f} scans {m[2]:.1f} | phase A {m[5]:.0f} us B {m[6]:.0f} us search {m[7]:.0f} us | bids {m[8]:.0f} | kernel {m[9]:.0f} us")
