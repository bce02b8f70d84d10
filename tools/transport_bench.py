"""Time cfm_transport_exact_f32 (exact OT between batches of different sizes; round 6: primal-dual phases with a tree
push) with and without the assignment warm start, and print its counters.
    python tools/transport_bench.py [B0xB1xd ...]
Measurement infrastructure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_amd.optimal_transport as ot
from cfm_amd import _lib

dev = torch.device("cuda", 0)
lib = _lib.load()
cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(127, 128, 2), (128, 127, 2), (125, 128, 2), (255, 256, 2), (200, 333, 16), (512, 500, 2),
                                                                        (511, 512, 2), (1000, 1001, 8), (1023, 1025, 3), (700, 1300, 4)]
for B0, B1, d in cases:
    g = torch.Generator().manual_seed(B0 * 7 + B1)
    x0 = torch.randn(B0, d, generator=g).to(dev); x1 = (torch.randn(B1, d, generator=g) * 0.7 + 0.5).to(dev)
    M = ot.cost_matrix(x0, x1)
    for warm in (False, True):
        ts = []
        try:
            for rep in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                plan, cost, info = ot.transport_exact(M, warm_start=warm, return_info=True)
                torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            print(f"{B0} x {B1} (d={d}) warm={int(warm)}: {1e3 * min(ts):9.3f} ms (incl. the warm start's square solve)  phases {info['phases']} sweeps {info['sweeps']} "
                  f"support {info['support']} staged {int(info['staged'])} warm used {int(info['warm_start_used'])}  cost {cost:.9f}", flush=True)
        except Exception as e:  # noqa: BLE001
            print(f"{B0} x {B1} (d={d}) warm={int(warm)}: FAILED {e!r}", flush=True)
