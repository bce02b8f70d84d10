import sys, time, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
lib = _lib.load(); dev = _lib.require_gpu()
rng = np.random.RandomState(0)
cases = {"allzero256": np.zeros((256, 256), np.float32), "binary256": rng.randint(0, 2, size=(256, 256)).astype(np.float32),
         "ties5_256": rng.randint(0, 5, size=(256, 256)).astype(np.float32), "dup256": np.repeat(rng.rand(256, 8).astype(np.float32), 32, axis=1),
         "identical_points": None, "rank1_256": np.outer(rng.rand(256), rng.rand(256)).astype(np.float32)}
x = rng.randn(1, 2).repeat(256, 0); y = rng.randn(256, 2)
cases["identical_points"] = ((x[:, None, :] - y[None]) ** 2).sum(-1).astype(np.float32)     # every row identical
import os
lib.cfm_assign_set_small(int(os.environ.get("SMALL", "1")))
for k, Mnp in cases.items():
    M = torch.from_numpy(Mnp).to(dev)
    perm, info = ot.assign_exact(M, return_info=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    perm, info = ot.assign_exact(M, return_info=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    p = perm.cpu().numpy().astype(np.int64)
    c, cr = oracle.assignment_cost(Mnp, p), oracle.assignment_cost(Mnp, oracle.exact_perm(Mnp))
    buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); b = list(buf)
    print(f"{k}: {dt:.2f} ms small={bool(info['stats'][7] & 0x40000000)} status={b[0]} rounds={info['stats'][0]} ok={c <= cr + 1e-9 * max(1, abs(cr))}")
