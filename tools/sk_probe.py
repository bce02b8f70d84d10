"""Run the Sinkhorn solvers at one benchmark shape for rocprofv3 (kernel trace / --pmc):
    python tools/sk_probe.py C5     matrix-streaming solver, B = 8192, d = 50, reg 0.1, 50 iterations
    python tools/sk_probe.py C2     variant B (on-the-fly cost) + matrix streaming, B = 4096, d = 2, reg 0.05
One config per process so that the per-kernel statistics are per config.  Measurement infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_amd  # noqa: F401
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
_lib.load(); dev = _lib.require_gpu()
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
reg = {"C2": 0.05, "C5": 0.1}[cfg]
x0, x1 = oracle.config_inputs(cfg)
a, b = x0.to(dev), x1.to(dev)
M = ot.cost_matrix(a, b)
for _ in range(2):
    ot.sinkhorn_log(M, reg, max_iter=50, stop_thr=0.0)
    if cfg == "C2":
        ot.sinkhorn_log_points(a, b, M, reg, max_iter=50, stop_thr=0.0)
torch.cuda.synchronize()
