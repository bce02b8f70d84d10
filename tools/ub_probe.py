"""Run the kernel-space entropic solvers (unbalanced, partial) at B = 4096 for rocprofv3 --kernel-trace: 45 iterations each
on the C2 clouds, reg = 5 (the float64 Gibbs kernel is alive there).  Measurement infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_amd  # noqa: F401
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
_lib.load(); dev = _lib.require_gpu()
a, b = oracle.config_inputs("C2")
M = ot.cost_matrix(a.to(dev), b.to(dev))
for _ in range(2):
    ot.unbalanced_plan(M, 5.0, 1.0, max_iter=45, stop_thr=0.0)
    ot.partial_plan(M, 5.0, 1.0, max_iter=45, stop_thr=0.0)
torch.cuda.synchronize()
