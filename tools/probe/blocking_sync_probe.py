"""Does a blocking event wait really sleep on this stack?  CPU time of the waiting thread during a ~100 ms device job, with
torch.cuda.Event(blocking=False / True) and with the library's own solver events (cfm_set_blocking_sync 0 / 1).
Measurement infrastructure (round 6)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from cfm_amd import _lib
import cfm_amd.optimal_transport as ot
lib = _lib.load(); dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev)
def job():
    x = a
    for _ in range(12): x = (x @ a) * 1e-4
    return x
job(); torch.cuda.synchronize()
for blocking in (False, True):
    ev = torch.cuda.Event(blocking=blocking)
    t0w, t0c = time.perf_counter(), time.thread_time()
    job(); ev.record(); ev.synchronize()
    print(f"torch Event(blocking={blocking}): wall {1e3 * (time.perf_counter() - t0w):.1f} ms, thread CPU {1e3 * (time.thread_time() - t0c):.1f} ms", flush=True)
g = torch.Generator().manual_seed(0)
Ms = [ot.cost_matrix(torch.randn(4096, 64, generator=g).to(dev), torch.randn(4096, 64, generator=g).to(dev)) for _ in range(8)]
with torch.cuda.stream(torch.cuda.Stream()):
    for flag in (0, 1, 0, 1):
        lib.cfm_set_blocking_sync(flag)
        ot.assign_exact_batch(Ms); torch.cuda.synchronize()
        t0w, t0c = time.perf_counter(), time.thread_time()
        for _ in range(10): ot.assign_exact_batch(Ms)
        print(f"cfm_set_blocking_sync({flag}): 10 batches of 8 solves: wall {1e3 * (time.perf_counter() - t0w):.1f} ms, thread CPU {1e3 * (time.thread_time() - t0c):.1f} ms", flush=True)
