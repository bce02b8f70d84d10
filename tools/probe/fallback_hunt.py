"""Hunt for the rare dense fallback of the exact solver under the pipelined loop's contention (round 6 soak: 1 in ~112 k
couplings): three worker threads solve batches of four C3 problems on their own streams while the main thread runs the
C3 model step; the process-wide fallback counter is read after every batch and the first hit is reported with the
device error code of the list path (9: more than 64 free rows handed over; 3 / 4 / 5: a broken forest; -1: certificate).
    python tools/probe/fallback_hunt.py [seconds] [B] [d]
Measurement infrastructure."""
import ctypes, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
import cfm_amd
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
import bench
lib = _lib.load(); dev = _lib.require_gpu()
T = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D = int(sys.argv[3]) if len(sys.argv) > 3 else 784
pool = bench.synth_batches(B, D, 16, 1000, dev)
stop = threading.Event(); hits = []; counts = [0, 0, 0]


def fb():
    b = (ctypes.c_int * 2)(); lib.cfm_assign_debug_fallback(b); return int(b[0]), int(b[1])


def worker(w):
    lib.cfm_set_blocking_sync(1)
    s = torch.cuda.Stream(device=dev)
    k = w
    with torch.cuda.stream(s):
        while not stop.is_set():
            grp = [pool[(k + q) % len(pool)] for q in range(4)]; k += 4
            Ms = [ot.cost_matrix(a, b) for a, b in grp]
            before = fb()[0]
            perms, infos = ot.assign_exact_batch(Ms, return_info=True) if "return_info" in ot.assign_exact_batch.__code__.co_varnames else (ot.assign_exact_batch(Ms), None)
            s.synchronize()
            counts[w] += 4
            after = fb()
            if after[0] != before:
                hits.append((w, counts[w], after, None if infos is None else [i.get("stats") for i in infos]))
                stop.set()


net = cfm_amd.MLP(dim=784, time_varying=True, w=512).to(dev)
opt = cfm_amd.FusedAdam(net.parameters(), lr=1e-4); reg = cfm_amd.RegressionStep(net, opt)
a, b = bench.synth_batches(4096, 784, 1, 7, dev)[0]; t = torch.rand(4096, device=dev)
ths = [threading.Thread(target=worker, args=(w,)) for w in range(3)]
for th in ths: th.start()
t0 = time.time(); steps = 0
while time.time() - t0 < T and not stop.is_set():
    for _ in range(8): reg(t, a, b)
    torch.cuda.current_stream().synchronize(); steps += 8
stop.set()
for th in ths: th.join()
print(f"{sum(counts)} solves in batches of four beside {steps} model steps, {time.time() - t0:.0f} s; fallback counter {fb()}")
for h in hits: print("HIT worker %d after %d solves: counter %s stats %s" % h)
