"""Exact-assignment report on the GPU: parity against SciPy on a zoo of instances, then the solve
time on C3 instances (B=4096, d=784) with the per-mode breakdown the device state machine books.

    python tools/asg_report.py [--seeds 2] [--zoo 1] [--sweep 1]

Test infrastructure (uses the oracle as the checker); not part of the product path."""
import argparse
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch

import cfm_amd  # noqa: F401
from cfm_amd import _lib
if os.environ.get("CFM_LIB_PATH"):          # a variant build of the library (tools/probe/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["CFM_LIB_PATH"])
import cfm_amd.optimal_transport as ot
import cfm_oracle as oracle
import bench

MODES = ["UMIN0", "INITRED", "AUCTION", "ARR", "CONVERT", "UMIN", "COLRED", "ROOTMIN", "SAP(relax)",
         "MS_FINISH", "CERT", "BUILD", "SOLVER", "DONE"]


def check(name, Mnp, dev, unique=True):
    M = torch.from_numpy(np.ascontiguousarray(Mnp, dtype=np.float32)).to(dev)
    t0 = time.perf_counter()
    try:
        perm, info = ot.assign_exact(M, return_info=True)
    except Exception as e:  # noqa: BLE001
        print(f"  FAIL {name}: {e}", flush=True)
        return False
    dt = time.perf_counter() - t0
    p = perm.cpu().numpy().astype(np.int64)
    n = len(p)
    ok = sorted(p.tolist()) == list(range(n))
    ref = oracle.exact_perm(Mnp)
    c, cr = oracle.assignment_cost(Mnp, p), oracle.assignment_cost(Mnp, ref)
    ok = ok and c <= cr + 1e-9 * max(1.0, abs(cr))
    if unique:
        ok = ok and np.array_equal(p, ref)
    fb = (ctypes.c_int * 2)(); _lib.load().cfm_assign_debug_fallback(fb)
    print(f"  {'ok  ' if ok else 'FAIL'} {name}: n={n} {dt*1e3:.1f} ms stats={info['stats']} fallbacks(total,last err)={fb[0]},{fb[1]}", flush=True)
    return ok


def zoo(dev):
    rng = np.random.RandomState(0)
    ok = True
    for n in (2, 3, 7, 64, 128, 257, 512, 1000, 1024, 2048):
        ok &= check(f"uniform n={n}", rng.rand(n, n).astype(np.float32) * 10, dev)
    for n, d in ((256, 2), (700, 2), (1024, 50), (2048, 784)):
        x = rng.randn(n, d); y = rng.randn(n, d) + 0.5
        ok &= check(f"geometric n={n} d={d}", ((x[:, None, :] - y[None]) ** 2).sum(-1).astype(np.float32) if n * n * d < 3e8
                    else (np.add.outer((x * x).sum(1), (y * y).sum(1)) - 2 * x @ y.T).astype(np.float32), dev)
    ok &= check("ties 0..4 n=200", rng.randint(0, 5, size=(200, 200)).astype(np.float32), dev, unique=False)
    ok &= check("all equal n=65", np.zeros((65, 65), dtype=np.float32), dev, unique=False)
    ok &= check("negative large n=90", -rng.rand(90, 90).astype(np.float32) * 1e4, dev)
    M = rng.rand(128, 64).astype(np.float32)
    ok &= check("duplicated columns n=128", np.repeat(M, 2, axis=1), dev, unique=False)
    ok &= check("uniform n=4096", rng.rand(4096, 4096).astype(np.float32), dev)
    print("ZOO", "ALL OK" if ok else "FAILURES", flush=True)
    return ok


def timing(dev, seeds, B=4096, d=784, label=""):
    lib = _lib.load()
    Ms = []
    for seed in range(1000, 1000 + 1000 * seeds, 1000):
        for (x0, x1) in bench.synth_batches(B, d, 8, seed, dev):
            Ms.append(ot.cost_matrix(x0, x1))
    ws = _lib.workspace(_lib.OP_ASSIGN, B, B, 0, dev)
    acc = np.zeros(32); evs = []; stats = []
    chk = 0
    for rep in range(2):
        for M in Ms:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); perm, info = ot.assign_exact(M, return_info=True); e1.record(); torch.cuda.synchronize()
            if rep == 1:
                chk = (chk * 1000003 + int((perm.long() * torch.arange(1, B + 1, device=dev)).sum().item())) % (2 ** 61 - 1)
                buf = (ctypes.c_double * 32)()
                _lib.check(lib.cfm_assign_debug_times(_lib.ptr(ws), buf), "dbg")
                acc += np.array(list(buf)); evs.append(e0.elapsed_time(e1) * 1e3); stats.append(info["stats"])
    # wall time of a solve without the statistics read-back (what a training step pays)
    walls = []
    for M in Ms:
        torch.cuda.synchronize(); t0 = time.perf_counter(); ot.assign_exact(M); walls.append((time.perf_counter() - t0) * 1e6)
    n = len(Ms)
    st = np.array(stats, dtype=np.float64).mean(0)
    fb = (ctypes.c_int * 2)(); lib.cfm_assign_debug_fallback(fb)
    print(f"[{label}] B={B} d={d}: {n} instances, perm checksum {chk}; dense-machine fallbacks so far {fb[0]} (last device error {fb[1]})")
    print(f"  mean solve {np.mean(evs):.0f} us (events, incl. read-back)  host wall w/o read-back {np.mean(walls):.0f} us "
          f"(min {np.min(walls):.0f} max {np.max(walls):.0f}); booked on the device {acc[:16].sum()/n:.0f} us")
    print(f"  stats mean: auction_rounds {st[0]:.1f} arr {st[1]:.1f} free_after_arr {st[2]:.1f} sap_batches {st[3]:.1f} "
          f"sap_scans {st[4]:.0f} total_scans {st[5]:.0f} steps {st[6]:.1f}")
    for q, nm in enumerate(MODES):
        if acc[q] > 0:
            print(f"    {nm:12s} {acc[q]/n:8.1f} us  {100*acc[q]/acc[:16].sum():5.1f} %")
    sys.stdout.flush()
    return float(np.mean(walls))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--zoo", type=int, default=1)
    ap.add_argument("--sweep", type=int, default=0)
    ap.add_argument("--sweep2", type=int, default=0, help="round-3 sweep: epsilon = 0 rounds, last epsilon, theta (forest phases in the list solver)")
    a = ap.parse_args()
    lib = _lib.load(); dev = _lib.require_gpu()
    if a.zoo:
        zoo(dev)
    with torch.cuda.stream(torch.cuda.Stream()):
        timing(dev, a.seeds, label="default")
        if a.sweep:
            for b in (80, 112, 128, 0):
                lib.cfm_assign_set_bulk(b, 0); timing(dev, 1, label=f"bulk {b}")
            lib.cfm_assign_set_bulk(96, 0)
            for c in (6, 16):
                lib.cfm_assign_set_params(0, 0, 0, -1, 0, -1, c); timing(dev, 1, label=f"chunk {c}")
            lib.cfm_assign_set_params(0, 0, 0, -1, 0, -1, 10)
            for h in (6, 16, 32):
                lib.cfm_assign_set_handoff(h); timing(dev, 1, label=f"handoff {h}")
            lib.cfm_assign_set_handoff(64)
            for th in (4.0, 7.0):
                lib.cfm_assign_set_params(th, 0, 0, -1, 0, -1, 0); timing(dev, 1, label=f"theta {th}")
            lib.cfm_assign_set_params(5.0, 0, 0, -1, 0, -1, 0)
            for ac in (10, 25):
                lib.cfm_assign_set_params(0, 0, 0, -1, 0, ac, 0); timing(dev, 1, label=f"arr_cap {ac}")
            lib.cfm_assign_set_params(0, 0, 0, -1, 0, 15, 0)
            lib.cfm_assign_set_mode(0); timing(dev, 1, label="dense only (no list solver)"); lib.cfm_assign_set_mode(1)
        if a.sweep2:
            for ac in (8, 10, 12):
                lib.cfm_assign_set_params(0, 0, 0, -1, 0, ac, 0); timing(dev, 1, label=f"arr_cap {ac}")
            lib.cfm_assign_set_params(0, 0, 0, -1, 0, 10, 0)
            return
            for el in (1e-5, 1e-4):
                lib.cfm_assign_set_params(0, 0, el, -1, 0, -1, 0); timing(dev, 1, label=f"eps_last {el} (arr_cap 10)")
            lib.cfm_assign_set_params(0, 0, 1e-6, -1, 0, -1, 0)
            for th in (7.0, 10.0):
                lib.cfm_assign_set_params(th, 0, 0, -1, 0, -1, 0); timing(dev, 1, label=f"theta {th} (arr_cap 10)")
            lib.cfm_assign_set_params(5.0, 0, 0, -1, 0, 15, 0)
            return
        timing(dev, 1, B=8192, d=50, label="B=8192 d=50")
        timing(dev, 1, B=1024, d=784, label="B=1024")
        timing(dev, 1, B=256, d=2, label="B=256 d=2")


if __name__ == "__main__":
    main()
