"""Cycle breakdown of the one-workgroup list solver (library built with -DSP_PROFILE, see tools/probe).
    CFM_PROF_LIB=tools/probe/libcfm_prof.so python tools/asg_solver_prof.py
Measurement infrastructure; not part of the product path."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cfm_amd  # noqa: F401
from cfm_amd import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("CFM_PROF_LIB", os.path.join(ROOT, "tools/probe/libcfm_prof.so")))
import cfm_amd.optimal_transport as ot
import bench
lib = _lib.load(); dev = _lib.require_gpu()
B = 4096
names = ["phase init (labels, roots)", "fast batches", "collect", "a-posteriori", "phase finish (accept, duals, augment)", "dense batches", "#fast", "#batches", "#phases",
         "fb: entries+lists issue", "fb: gathers+lower", "fb: barrier1", "fb: phase W", "fb: barrier2", "fb: radius (last wave)", "pending seen"]
with torch.cuda.stream(torch.cuda.Stream()):
    Ms = [ot.cost_matrix(x0, x1) for (x0, x1) in bench.synth_batches(B, 784, 8, 1000, dev) + bench.synth_batches(B, 784, 8, 2000, dev)]
    ws = _lib.workspace(_lib.OP_ASSIGN, B, B, 0, dev)
    for asy in [int(x) for x in os.environ.get("ASYNC_LIST", "0,2").split(",")]:
        lib.cfm_assign_set_async(asy, -1, -1)
        print(f"== cfm_assign_set_async({asy}): {'synchronous bid rounds' if asy == 0 else 'one-launch asynchronous auction'}")
        acc = np.zeros(16); free = []
        for M in Ms:
            _, info = ot.assign_exact(M, return_info=True); torch.cuda.synchronize()
            free.append(info["stats"][2])
            buf = (ctypes.c_longlong * 16)()
            _lib.check(lib.cfm_assign_debug_solver(_lib.ptr(ws), B, buf), "dbg")
            acc += np.array(list(buf), dtype=np.float64)
        acc /= len(Ms)
        print(f"mean per solve over {len(Ms)} instances, free rows handed to the solver {np.mean(free):.1f} (cycles of clock64; per batch in parentheses):")
        nb = max(acc[7], 1)
        for q, nm in enumerate(names):
            print(f"  {nm:26s} {acc[q]:12.0f}   ({acc[q]/nb:9.1f})")
