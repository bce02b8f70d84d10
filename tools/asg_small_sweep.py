"""Sweep the epsilon schedule of the one-workgroup solver (n <= 256) over a few instances: total time per instance set.
    python tools/asg_small_sweep.py
Measurement infrastructure."""
import sys, time, ctypes, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
rng = np.random.RandomState(0)
inst = []
x0, x1 = oracle.config_inputs("C1")
inst.append(("C1", ot.cost_matrix(x0.to(dev), x1.to(dev), matrix_cores=False)))
for seed in range(3):
    a, b = oracle.config_inputs("C1", rank=seed + 1)
    inst.append((f"C1r{seed+1}", ot.cost_matrix(a.to(dev), b.to(dev), matrix_cores=False)))
for n, d in ((256, 784), (128, 2), (128, 784), (64, 2)):
    x = torch.from_numpy(rng.randn(n, d).astype(np.float32)).to(dev); y = torch.from_numpy((rng.randn(n, d) + 0.5).astype(np.float32)).to(dev)
    inst.append((f"n{n}d{d}", ot.cost_matrix(x, y, matrix_cores=False)))
inst.append(("uni256", torch.from_numpy((rng.rand(256, 256) * 10).astype(np.float32)).to(dev)))
ref = {k: ot.assign_exact(M).cpu() for k, M in inst}


def timeit(M):
    for _ in range(2): ot.assign_exact(M)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): ot.assign_exact(M)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 30 * 1e6


def stats():
    buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); return list(buf)


print("config: " + " ".join(k for k, _ in inst) + " | sum | free rows per instance")
grid = [(th, 8e-3, el, 0.02, ac) for th in (2.0, 2.5, 3.0) for el in (1e-6, 5e-7) for ac in (15, 30)]
grid += [(2.5, e0, 1e-6, sf, 15) for e0 in (4e-3, 1.6e-2) for sf in (0.02, 0.03)]
for theta, eps0, el, sf, ac in grid:
    lib.cfm_assign_set_params(theta, eps0, el, sf, 0, ac, 0)
    ts, fr = [], []
    for k, M in inst:
        ts.append(timeit(M)); fr.append(stats()[2])
        assert torch.equal(ot.assign_exact(M).cpu(), ref[k]), (k, theta, eps0)
    print(f"theta {theta:3.0f} eps0 {eps0:.0e} last {el:.0e} stop {sf:.2f} arr {ac:2d}: " + " ".join(f"{t:5.0f}" for t in ts) + f" | {sum(ts):6.0f} | " + " ".join(str(f) for f in fr))
