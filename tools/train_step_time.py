"""Model-step time at C3 (B=4096, 785-512-512-512-784): HIP kernels + FusedAdam vs PyTorch-ROCm + torch Adam.
Measurement infrastructure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch, cfm_amd
from cfm_amd import _lib
_lib.load(); dev = _lib.require_gpu()
B, d = 4096, 784
xt = torch.randn(B, d, device=dev); t = torch.rand(B, device=dev); ut = torch.randn(B, d, device=dev)
for hip, fused in ((True, True), (False, False), (True, False), (False, True)):
    torch.manual_seed(0)
    m = cfm_amd.MLP(dim=d, time_varying=True, w=512).to(dev); m.hip_training = hip
    opt = cfm_amd.FusedAdam(m.parameters()) if fused else torch.optim.Adam(m.parameters())
    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.mean((m(torch.cat([xt, t[:, None]], dim=-1)) - ut) ** 2)
        loss.backward(); opt.step()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 50
    for _ in range(n): step()
    torch.cuda.synchronize()
    print(f"hip_training={hip} fused_adam={fused}: {(time.perf_counter()-t0)/n*1e3:.3f} ms / model step", flush=True)
