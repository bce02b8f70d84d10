import sys, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
lib = _lib.load(); dev = _lib.require_gpu()
rng = np.random.RandomState(1)
for n in (2, 3, 4, 5, 7, 8, 9, 16, 31, 33):
    for rep in range(3):
        M = torch.from_numpy((rng.rand(n, n) * 10).astype(np.float32)).to(dev)
        perm, info = ot.assign_exact(M, return_info=True)
        buf = (ctypes.c_int * 16)(); lib.cfm_assign_debug_small(buf); b = list(buf)
        print(n, "status", b[0], "phases", b[10], "cut", b[12], "bids", b[9], "free", b[2], "small" if info["stats"][7] & 0x40000000 else "FALLBACK")
