"""Sweep of the exact solver's epsilon schedule / hand-over point at C3 size (round 6: VERDICT r5 #1a — hand the list
solver fewer rows): per configuration the mean lone-solve time over NINST instances, the time booked per mode, the free
rows handed over, row evaluations; permutations must be identical in every configuration.
    python tools/asg_sched_sweep.py "theta=2.5" "theta=2.5,arr=20" "last_div=16,eps_last=1e-7" ...
keys: theta eps0 eps_last stop arr last_div stop_early handoff blocks  (unset keys = the shipped defaults)
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
DEF = dict(theta=2.5, eps0=8e-3, eps_last=1e-6, stop=0.02, arr=10, last_div=4, stop_early=0.0, handoff=64, blocks=16)
NI = int(os.environ.get("NINST", "16"))
Ms = []
for k in range(NI):
    x0, x1 = oracle.config_inputs("C3", rank=k)
    Ms.append(ot.cost_matrix(x0.to(dev), x1.to(dev)))
if os.environ.get("BENCH_POOL"):
    import bench
    for seed in (1000, 2000, 3000):
        Ms += [ot.cost_matrix(a, b) for (a, b) in bench.synth_batches(4096, 784, 8, seed, dev)]
B = Ms[0].shape[0]
MODES = ["umin0", "initred", "auction", "arr", "convert", "umin", "colred", "rootmin", "sap", "ms_finish", "cert", "build", "solver"]


def apply(cfg):
    c = dict(DEF); c.update(cfg)
    lib.cfm_assign_set_params(c["theta"], c["eps0"], c["eps_last"], c["stop"], 0, int(c["arr"]), 0)
    lib.cfm_assign_set_async(2, int(c["blocks"]), int(c["last_div"]))
    lib.cfm_assign_set_stop_early(c["stop_early"])
    lib.cfm_assign_set_handoff(int(c["handoff"]))


perms = {}
specs = sys.argv[1:] or ["theta=2.5"]
with torch.cuda.stream(torch.cuda.Stream()):
    ws = _lib.workspace(_lib.OP_ASSIGN, B, B, 0, dev)
    for spec in specs:
        cfg = {k: float(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}
        apply(cfg)
        for M in Ms[:3]:
            ot.assign_exact(M)
        ts, acc, st, adopt = [], np.zeros(16), [], []
        fb0 = (ctypes.c_int * 2)(); lib.cfm_assign_debug_fallback(fb0)
        for rep in range(2):
            for q, M in enumerate(Ms):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                perm, info = ot.assign_exact(M, return_info=True); torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
                buf = (ctypes.c_double * 32)(); lib.cfm_assign_debug_times(_lib.ptr(ws), buf)
                acc += np.array(list(buf))[:16]; st.append(info["stats"]); adopt.append((buf[17], buf[18]))
                perms.setdefault(q, perm.cpu()); assert torch.equal(perms[q], perm.cpu()), (spec, q)
        fb1 = (ctypes.c_int * 2)(); lib.cfm_assign_debug_fallback(fb1)
        sta = np.array(st, dtype=float)
        stm = sta.mean(0)
        nn = len(ts)
        # batch of 4 (the throughput form)
        tb = []
        ot.assign_exact_batch(Ms[:4]); torch.cuda.synchronize()
        for g0 in range(0, len(Ms) - 3, 4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ot.assign_exact_batch(Ms[g0:g0 + 4]); torch.cuda.synchronize(); tb.append(time.perf_counter() - t0)
        print(f"{spec:44s} lone mean {1e3 * np.mean(ts):.3f} med {1e3 * np.median(ts):.3f} max {1e3 * max(ts):.3f} ms | free rows {stm[2]:.1f} (max {sta[:, 2].max():.0f}) "
              f"evals {stm[5]:.0f} batches {stm[3]:.0f} phases {((sta[:, 7].astype(int) >> 8) & 0xff).mean():.1f} | us: auction {acc[2] / nn:.0f} build {acc[11] / nn:.0f} solver {acc[12] / nn:.0f} other {(acc[:11].sum() - acc[2]) / nn:.0f} | "
              f"batch4 {1e3 * np.median(tb) / 4:.3f} ms/problem | fallbacks {fb1[0] - fb0[0]} | groups adopted {np.mean([a for a, _ in adopt]):.1f} claims {np.mean([c for _, c in adopt]):.0f}", flush=True)
apply({})
print("permutations identical in every configuration")
