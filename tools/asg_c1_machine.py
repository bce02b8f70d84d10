"""C1-sized problems (n = 64 .. 512) on the chip-wide machine with the one-launch auction, against the one-workgroup solver
(assign_small.h): does the asynchronous auction make the machine the faster path at tutorial sizes?  (round 6)
Measurement infrastructure."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
from cfm_amd import _lib
lib = _lib.load(); dev = _lib.require_gpu()
rng = np.random.RandomState(0)
cases = []
x0, x1 = oracle.config_inputs("C1")
cases.append(("C1 256 d=2", ot.cost_matrix(x0.to(dev), x1.to(dev))))
for n, d in ((256, 784), (128, 2), (512, 2), (200, 16)):
    x = torch.from_numpy(rng.randn(n, d).astype(np.float32)).to(dev); y = torch.from_numpy((rng.randn(n, d) + 0.5).astype(np.float32)).to(dev)
    cases.append((f"n={n} d={d}", ot.cost_matrix(x, y)))
def run(tag):
    for name, M in cases:
        for _ in range(3): ot.assign_exact(M)
        torch.cuda.synchronize(); ts = []
        for _ in range(20):
            t0 = time.perf_counter(); p, info = ot.assign_exact(M, return_info=True); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ref = oracle.exact_perm(M.cpu().numpy())
        ok = np.array_equal(p.cpu().numpy(), ref)
        print(f"{tag:34s} {name:14s} median {1e6 * np.median(ts):7.0f} us  min {1e6 * min(ts):7.0f}  launches {info['stats'][6]}  free rows {info['stats'][2]}  == scipy {ok}", flush=True)
with torch.cuda.stream(torch.cuda.Stream()):
    run("one-workgroup solver (default)")
    lib.cfm_assign_set_small(0)
    run("machine, synchronous rounds")
    lib.cfm_assign_set_async_min_n(64); lib.cfm_assign_set_bulk(96, 64)       # (the whole solve as the unpolled head from n = 64)
    run("machine, one-launch auction")
    lib.cfm_assign_set_params(2.5, 0, 0, 0.05, 0, -1, 0)
    run("  ... stop_frac 0.05")
    lib.cfm_assign_set_params(4.0, 0, 0, 0.05, 0, -1, 0)
    run("  ... theta 4")
