"""Time cfm_transport_exact_f32 (exact OT between batches of different sizes) and print its counters.
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
cases = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(127, 128, 2), (255, 256, 2), (200, 333, 16), (512, 500, 2)]
for B0, B1, d in cases:
    g = torch.Generator().manual_seed(B0 * 7 + B1)
    x0 = torch.randn(B0, d, generator=g).to(dev); x1 = (torch.randn(B1, d, generator=g) * 0.7 + 0.5).to(dev)
    M = ot.cost_matrix(x0, x1)
    plan = torch.empty((B0, B1), dtype=torch.float64, device=dev); tot = torch.empty(1, dtype=torch.float64, device=dev)
    info = torch.empty(8, dtype=torch.int32, device=dev)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        _lib.check(lib.cfm_transport_exact_f32(_lib.ptr(M), B0, B1, _lib.ptr(plan), _lib.ptr(tot), _lib.ptr(info), _lib.stream_ptr()), "tp")
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = info.cpu().tolist()
    print(f"{B0} x {B1} (d={d}): {1e3*min(ts):9.2f} ms  status {st[0]} searches {st[1]} row relaxations {st[2]} support {st[3]} "
          f"units {st[5]}/{st[6]} matrix in LDS {st[7]}  cost {float(tot.cpu()[0]):.9f}")
