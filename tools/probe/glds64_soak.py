"""Determinism soak of the 64 x 64 direct-to-LDS layer engine (gemm_glds64.h): the C3 forward and the fused regression step repeated under
contention (a second thread solving C3 assignment problems on its own stream), every result compared bit for bit with the first.
Measurement infrastructure."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
lib = _lib.load(); dev = _lib.require_gpu(); torch.manual_seed(0)
T = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
stop = threading.Event()
def noise():
    s = torch.cuda.Stream(device=dev); pool = bench.synth_batches(4096, 784, 4, 1000, dev)
    with torch.cuda.stream(s):
        k = 0
        while not stop.is_set():
            a, b = pool[k % 4]; k += 1
            ot.assign_exact(ot.cost_matrix(a, b)); s.synchronize()
th = threading.Thread(target=noise); th.start()
bad = 0; n = 0
for (B, d, w) in ((4096, 784, 512), (1000, 48, 64), (257, 32, 512)):
    net = cfm_amd.MLP(dim=d, time_varying=True, w=w).to(dev)
    x = torch.randn(B, d, device=dev); t = torch.rand(B, device=dev); u = torch.randn(B, d, device=dev)
    opt = cfm_amd.FusedAdam(net.parameters(), lr=0.0); reg = cfm_amd.RegressionStep(net, opt)
    with torch.no_grad(): ref = net.forward_hip(x, t).clone()
    l0 = float(reg.backward_only(t, x, u)); g0 = [p.grad.clone() for p in net.parameters()]
    t0 = time.time()
    while time.time() - t0 < T / 3:
        with torch.no_grad(): o = net.forward_hip(x, t)
        l = float(reg.backward_only(t, x, u))
        ok = torch.equal(o, ref) and l == l0 and all(torch.equal(a, p.grad) for a, p in zip(g0, net.parameters()))
        bad += 0 if ok else 1; n += 1
stop.set(); th.join()
print(f"{n} forward + regression-step repetitions at three shapes under contention: {bad} differ from the first")
