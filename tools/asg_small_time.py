import sys, time, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
rng = np.random.RandomState(0)
def run(name, M):
    for _ in range(3): ot.assign_exact(M)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): ot.assign_exact(M)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e6
    buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); b = list(buf)
    print(f"{name}: {dt:.0f} us status {b[0]} | looks {b[1]} bids {b[9]} phases {b[10]} cut {b[12]} searches {b[13]} search scans {b[14]} arr {b[11]} free {b[2]} scans {b[3]} | phase A {b[15]/100:.0f} us | load+init {b[4]/100:.0f} bid {b[5]/100:.0f} convert {b[6]/100:.0f} search {b[7]/100:.0f} cert {b[8]/100:.0f} us")
x0, x1 = oracle.config_inputs("C1")
run("C1 (256, d=2)", ot.cost_matrix(x0.to(dev), x1.to(dev), matrix_cores=False))
for n, d in ((256, 784), (128, 2), (128, 784), (64, 2)):
    x = torch.from_numpy(rng.randn(n, d).astype(np.float32)).to(dev); y = torch.from_numpy((rng.randn(n, d) + 0.5).astype(np.float32)).to(dev)
    run(f"n={n} d={d}", ot.cost_matrix(x, y, matrix_cores=False))
run("uniform 256", torch.from_numpy((rng.rand(256, 256) * 10).astype(np.float32)).to(dev))
