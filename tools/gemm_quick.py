"""Quick A/B timing of the dense products at the C3 shapes (no correctness checks: tools/gemm_bench.py has them):
cost matrix, inference forward, the fused regression step (forward + MSE + backward) and Adam.  Median of 5 x 30 calls.
    CFM_LIB_PATH=tools/probe/libcfm_X.so python tools/gemm_quick.py
Measurement infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
_lib.load(); dev = _lib.require_gpu(); torch.manual_seed(0)
B, d = 4096, 784


def med(fn, reps=30, rounds=5):
    fn(); fn(); torch.cuda.synchronize(); out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize(); out.append(e0.elapsed_time(e1) / reps * 1e3)
    return float(np.median(out))


a = torch.randn(B, d, device=dev); b = torch.clamp(0.35 * torch.randn(B, d, device=dev) + 0.2, -1, 1)
t = torch.rand(B, device=dev)
net = cfm_amd.MLP(dim=d, time_varying=True, w=512).to(dev)
opt = cfm_amd.FusedAdam(net.parameters(), lr=1e-4); reg = cfm_amd.RegressionStep(net, opt)
us_cost = med(lambda: ot.cost_matrix(a, b))
with torch.no_grad():
    us_fwd = med(lambda: net.forward_hip(a, t))
us_fb = med(lambda: reg.backward_only(t, a, b))
us_step = med(lambda: reg(t, a, b))
print(f"{os.environ.get('CFM_LIB_PATH', 'main'):40s} cost {us_cost:7.1f} us ({2.0*B*B*d/us_cost/1e6:5.1f} TF)  fwd {us_fwd:6.1f} us ({10.88e3/us_fwd:5.1f} TF)  "
      f"fwd+mse+bwd {us_fb:6.1f} us ({32.6e3/us_fb:5.1f} TF)  step {us_step:6.1f} us")
