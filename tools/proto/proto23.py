"""Prototype: the auction's schedule knobs on the CPU — rounds and rows left free after the epsilon = 0 rounds, per
(theta, stop_frac, stop_early, eps0, eps_last, arr rounds), at C3.  Same emulation as proto21 (Jacobi rounds, one winner
per object, all rows unassigned at a phase start, prices kept), plus the epsilon = 0 rounds (no reset).  The device pays
~8.2 us per round and ~45 us per row left free (list solver), so cost ~ 8.2 * rounds + 45 * free."""
import sys, os, time, itertools
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cfm_oracle as oracle


def rounds_of(C, theta=5.0, eps0_frac=8e-3, eps_last_frac=1e-6, stop_frac=0.02, stop_early=0.0, arr=10):
    n = C.shape[0]
    rng = C.max() - C.min()
    u = C.min(1)
    p = (u[:, None] - C).max(0)
    eps = eps0_frac * rng
    owner = np.full(n, -1); a = np.full(n, -1)
    rounds = 0; bids = 0

    def one_round(free, eps):
        v = C[free] + p
        oo = np.argpartition(v, 1, axis=1)[:, :2]
        v2 = np.take_along_axis(v, oo, 1)
        sw = v2[:, 0] > v2[:, 1]
        bj = np.where(sw, oo[:, 1], oo[:, 0]); bb = v2.min(1); ss = v2.max(1)
        bid = p[bj] + (ss - bb) + eps
        order = np.lexsort((free, bid))
        win = {}
        for k in order: win[bj[k]] = k
        for j, k in win.items():
            i = free[k]
            if bid[k] > p[j] or owner[j] < 0 or (eps == 0.0 and bid[k] >= p[j]):
                if owner[j] >= 0: a[owner[j]] = -1
                owner[j] = i; a[i] = j; p[j] = max(p[j], bid[k])
    while True:
        is_last = (eps / theta) < eps_last_frac * rng
        stop = int((stop_frac if is_last else max(stop_frac, stop_early)) * n)
        owner[:] = -1; a[:] = -1
        first = True
        while True:
            free = np.where(a < 0)[0]
            if len(free) == 0 or (not first and len(free) <= stop): break
            rounds += 1; bids += len(free); one_round(free, eps); first = False
        if is_last: break
        eps /= theta
    free_before = int((a < 0).sum())
    for _ in range(arr):
        free = np.where(a < 0)[0]
        if len(free) == 0: break
        rounds += 1; bids += len(free); one_round(free, 0.0)
    return rounds, bids, free_before, int((a < 0).sum())


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    seeds = [0, 1] if len(sys.argv) < 3 else [int(s) for s in sys.argv[2].split(",")]
    Cs = []
    for r in seeds:
        x0, x1 = oracle.config_inputs("C3", B=B, rank=r)
        Cs.append(oracle.ref_cost_f32(x0, x1).astype(np.float64))
    grid = [dict()] + [dict(stop_early=s) for s in (0.05, 0.1, 0.2)] + [dict(stop_frac=s) for s in (0.01, 0.03, 0.05)] + \
           [dict(theta=t) for t in (4.0, 7.0, 10.0)] + [dict(eps0_frac=e) for e in (4e-3, 2e-2)] + \
           [dict(theta=7.0, stop_early=0.1), dict(theta=10.0, stop_early=0.1), dict(stop_frac=0.01, stop_early=0.1),
            dict(arr=15), dict(arr=20), dict(stop_frac=0.01, arr=15)]
    for kw in grid:
        res = np.array([rounds_of(C, **kw) for C in Cs], dtype=float).mean(0)
        cost = 8.2 * res[0] + 45.0 * res[3]
        print(f"{str(kw):50s} rounds {res[0]:6.1f} bids {res[1]:8.0f} free before/after eps=0 rounds {res[2]:6.1f} {res[3]:6.1f}   ~{cost:6.0f} us")
