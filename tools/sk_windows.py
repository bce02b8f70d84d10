"""Per-window view of a Sinkhorn bench leg (VERDICT r4 Next #2): for one config, N windows of `iters` iterations each
with HIP events, and for every window the solver's own state after it (iterations done, whether the fp64-exp regime
engaged, the last marginal violation) — plus the violation / regime as a function of the iteration count, so a slow
window can be told from a regime switch.
    python tools/sk_windows.py C5 [iters=200] [windows=7]
Measurement infrastructure."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import cfm_amd  # noqa: F401
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
_lib.load(); dev = _lib.require_gpu()
cfg = sys.argv[1] if len(sys.argv) > 1 else "C5"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nwin = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reg = {"C2": 0.05, "C5": 0.1}[cfg]
x0, x1 = oracle.config_inputs(cfg)
M = ot.cost_matrix(x0.to(dev), x1.to(dev))
B0, B1 = M.shape


def state(r):
    s = r.ws[:48].cpu()
    i = s[:16].view(torch.int32); d = s[16:48].view(torch.float64)
    return {"done": int(i[0]), "iters_done": int(i[1]), "precise": int(i[3]), "last_err": float(d[2])}


with torch.cuda.stream(torch.cuda.Stream()):
    ot.sinkhorn_log(M, reg, max_iter=20, stop_thr=0.0); torch.cuda.synchronize()
    for w in range(nwin):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ot.sinkhorn_log(M, reg, max_iter=iters, stop_thr=0.0); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); st = state(r)
        print(f"window {w}: {ms:8.3f} ms  {iters / ms * 1e3:9.0f} it/s  {(2 * 4.0 * B0 * B1 + 16 * B0) * iters / ms / 1e6:7.0f} GB/s  {st}")
    print("regime vs iteration count (each a fresh solve):")
    for n in (10, 20, 40, 60, 80, 100, 150, 200, 300, 400):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = ot.sinkhorn_log(M, reg, max_iter=n, stop_thr=0.0); e1.record(); torch.cuda.synchronize()
        print(f"  max_iter {n:4d}: {e0.elapsed_time(e1):8.3f} ms  {state(r)}")
