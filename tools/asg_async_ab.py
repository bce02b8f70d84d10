"""A/B of the asynchronous phase A (asg_auction) against the synchronous bid rounds at C3 size: per solve / per batch
time, launches, row evaluations, time booked per mode, and that the permutations are identical.
    python tools/asg_async_ab.py [nb ...]
Measurement infrastructure."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
from cfm_amd import _lib

lib = _lib.load(); dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:]] or [1, 4]
Ms = []
NI = int(os.environ.get("NINST", "8"))
for k in range(NI):
    x0, x1 = oracle.config_inputs("C3", rank=k)
    Ms.append(ot.cost_matrix(x0.to(dev), x1.to(dev)))
if os.environ.get("BENCH_POOL"):                     # bench.py's synthetic minibatches as well
    sys.path.insert(0, ROOT)
    import bench
    for seed in (1000, 2000, 3000):
        Ms += [ot.cost_matrix(a, b) for (a, b) in bench.synth_batches(4096, 784, 8, seed, dev)]
B = Ms[0].shape[0]
MODES = ["umin0", "initred", "auction", "arr", "convert", "umin", "colred", "rootmin", "sap", "ms_finish", "cert", "build", "solver"]
perms = {}
with torch.cuda.stream(torch.cuda.Stream()):
    ws = _lib.workspace(_lib.OP_ASSIGN, B, B, 0, dev)      # (the cached workspace of THIS stream: the one assign_exact uses)
    for asy in [tuple(int(y) for y in x.split(',')) for x in os.environ.get('ASYNC_LIST', '0,0,1;1,0,1;0,0,1;1,0,1').split(';')]:
        lib.cfm_assign_set_async(*asy[:3])
        if len(asy) > 3:                                       # 4th field: number of epsilon = 0 rounds
            lib.cfm_assign_set_params(0, 0, 0, -1, 0, asy[3], 0)
        sched = os.environ.get("SCHED")                        # "theta,eps0,eps_last,stop_frac" (0 / -1 keep the default)
        if sched:
            th, e0, el, sf = (float(x) for x in sched.split(","))
            lib.cfm_assign_set_params(th, e0, el, sf, 0, -1, 0)
        for M in Ms[:2]:
            ot.assign_exact(M)
        ts, acc, st = [], np.zeros(16), []
        for q, M in enumerate(Ms):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            perm, info = ot.assign_exact(M, return_info=True); torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            buf = (ctypes.c_double * 32)(); lib.cfm_assign_debug_times(_lib.ptr(ws), buf)
            acc += np.array(list(buf))[:16]; st.append(info["stats"])
            perms.setdefault(q, perm.cpu()); assert torch.equal(perms[q], perm.cpu()), (asy, q)
        st = np.array(st, dtype=float).mean(0)
        print(f"async={asy}  lone solve over {len(Ms)} instances: mean {1e3 * np.mean(ts):.3f}  median {1e3 * np.median(ts):.3f} ms  (min {1e3 * min(ts):.3f})  launches {st[6]:.1f}  row evaluations {st[5]:.0f}  "
              f"auction rounds {st[0]:.1f}  eps=0 rounds {st[1]:.1f}  free rows after {st[2]:.1f}")
        print("      us per mode: " + "  ".join(f"{m} {acc[i] / len(Ms):.0f}" for i, m in enumerate(MODES) if acc[i] > 0))
        for nb in sizes:
            if nb < 2:
                continue
            out = ot.assign_exact_batch(Ms[:nb]); torch.cuda.synchronize()
            for q in range(nb):
                assert torch.equal(out[q].cpu(), perms[q]), ("batch", asy, q)
            tb = []
            for g0 in range(0, len(Ms) - nb + 1, nb):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ot.assign_exact_batch(Ms[g0:g0 + nb]); torch.cuda.synchronize(); tb.append(time.perf_counter() - t0)
            print(f"      batch of {nb}: mean {1e3 * np.mean(tb):.3f}  median {1e3 * np.median(tb):.3f} ms = {1e3 * np.median(tb) / nb:.3f} ms per problem")
print("permutations identical in every configuration")
