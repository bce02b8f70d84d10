"""Per-instance view of the exact solver over bench.py's pool: lone solve and batch-of-4 times, launches, free rows handed to
the list solver, dense fallbacks — to find the instances behind slow regions of the pipelined loop.
    python tools/asg_pool_diag.py [pool size = 16]
Measurement infrastructure."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import cfm_amd.optimal_transport as ot
from cfm_amd import _lib
import bench
lib = _lib.load(); dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with torch.cuda.stream(torch.cuda.Stream()):
    Ms = [ot.cost_matrix(a, b) for (a, b) in bench.synth_batches(4096, 784, N, 1000, dev)]
    ot.assign_exact(Ms[0]); ot.assign_exact_batch(Ms[:4]); ot.assign_exact_batch(Ms[:2]); ot.assign_exact_batch(Ms[:1]); torch.cuda.synchronize()
    fb = (ctypes.c_int * 2)()
    for q, M in enumerate(Ms):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); perm, info = ot.assign_exact(M, return_info=True); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        st = info["stats"]; lib.cfm_assign_debug_fallback(fb)
        print(f"instance {q:2d}: lone {1e3 * min(ts):.3f} / {1e3 * max(ts):.3f} ms  launches {st[6]}  free rows {st[2]}  row evals {st[5]}  ms-phases {(st[7] >> 8) & 0xff}  dense fallbacks {st[7] >> 16}  redone {fb[0]} err {fb[1]}")
    for nb in (4, 2, 1):
        for g0 in range(0, N - nb + 1, nb):
            ts = []
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter(); perms, infos = ot.assign_exact_batch(Ms[g0:g0 + nb], return_info=True); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            lib.cfm_assign_debug_fallback(fb)
            print(f"batch of {nb} [{g0}:{g0 + nb}]: {1e3 * min(ts):.3f} / {1e3 * max(ts):.3f} ms  launches {[i['stats'][6] for i in infos]}  free rows {[i['stats'][2] for i in infos]}  redone {fb[0]}")
