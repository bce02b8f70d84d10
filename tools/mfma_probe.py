"""Run the MFMA-bearing kernels of the path a few times each (for rocprofv3 --pmc / --kernel-trace):
mlp_* (C3 field forward, B=4096), cost_gemm (C3 cost matrix), ode_small_dopri (C5 dopri5), the
layer-per-kernel ODE on the C3 field, gemm_f32_mfma (C3 model step: forward, backward, Adam).  Measurement infrastructure; not part of the product path."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
from cfm_amd.ode import NeuralODE
from cfm_amd.utils import torch_wrapper
import cfm_oracle as oracle
_lib.load(); dev = _lib.require_gpu()
torch.manual_seed(0)
x0, x1 = oracle.config_inputs("C3")
a, b = x0.to(dev), x1.to(dev)
m = cfm_amd.MLP(dim=784, time_varying=True, w=512).to(dev)
t = torch.rand(4096, device=dev)
with torch.no_grad():
    for _ in range(5):
        m.forward_hip(a, t)
        ot.cost_matrix(a, b, matrix_cores=True)
        ot.cost_matrix(a, b, matrix_cores=False)
    node = NeuralODE(torch_wrapper(m), solver="dopri5", atol=1e-4, rtol=1e-4)
    node.trajectory(a, torch.linspace(0, 1, 5))        # (B = 4096 like every other call: the per-kernel counters then describe ONE shape family, the C3 layers)
    c0, _ = oracle.config_inputs("C5")
    ms = cfm_amd.MLP(dim=50, time_varying=True, w=64).to(dev)
    nodes = NeuralODE(torch_wrapper(ms), solver="dopri5", atol=1e-4, rtol=1e-4)
    for _ in range(2):
        nodes.trajectory(c0.to(dev), torch.linspace(0, 1, 100))
# the training step of the C3 model as the bench runs it: cfm_amd.RegressionStep (forward with the time column fused, MSE,
# backward GEMMs, one reduction) + the one-launch Adam
mt = cfm_amd.MLP(dim=784, time_varying=True, w=512).to(dev)
opt = cfm_amd.FusedAdam(mt.parameters(), lr=1e-4)
reg = cfm_amd.RegressionStep(mt, opt)
for _ in range(8):
    reg(t, a, b)
torch.cuda.synchronize()
