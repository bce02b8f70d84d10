"""Profiling build of the one-workgroup solver (CFM_EXTRA_FLAGS=-DSMA_PROF): cycles between the marks of wave 0's iterations."""
import sys, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
x0, x1 = oracle.config_inputs("C1")
M = ot.cost_matrix(x0.to(dev), x1.to(dev), matrix_cores=False)
for _ in range(3):
    ot.assign_exact(M); torch.cuda.synchronize()
    buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); b = list(buf)
    print("looks", b[1], "bids", b[9], "| cycles: top->loads", b[8], "loads->pick", b[4], "pick", b[5], "eval", b[6], "bid", b[7], "| marks0", b[14], "marks3", b[15])
