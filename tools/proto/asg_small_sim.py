"""NumPy emulation of the planned one-workgroup integer auction (n <= 256): Jacobi rounds, eps scaling with
violators-only phase transitions, integer costs scaled by (n+1).  Counts rounds / bids; checks against SciPy."""
import sys, numpy as np
from scipy.optimize import linear_sum_assignment

def solve(M, theta=5, eps0_frac=2.0**-7, restart=False, verbose=False):
    n = M.shape[0]
    cmin, cmax = float(M.min()), float(M.max())
    rng = cmax - cmin
    if rng == 0: rng = 1.0
    S = (n + 1) * 2.0**43 / rng
    C = np.rint((M.astype(np.float64) - cmin) * S).astype(np.int64)
    P = np.zeros(n, np.int64); owner = -np.ones(n, np.int64); arow = -np.ones(n, np.int64)
    eps = max(int(C.max() * eps0_frac), 1)
    rounds = 0; bids = 0; phases = 0; log = []
    while True:
        phases += 1; r0 = rounds; b0 = bids
        while True:
            free = np.nonzero(arow < 0)[0]
            if len(free) == 0: break
            rounds += 1; bids += len(free)
            W = C[free] + P[None, :]
            j1 = W.argmin(1); w1 = W[np.arange(len(free)), j1]
            W2 = W.copy(); W2[np.arange(len(free)), j1] = np.iinfo(np.int64).max
            w2 = W2.min(1) if n > 1 else w1
            newP = P[j1] + (w2 - w1) + eps
            # column takes the max bid (ties: larger row id)
            order = np.lexsort((free, newP))
            for k in order:      # ascending: the last write per column wins = max
                j = j1[k]
                if newP[k] > P[j] or (newP[k] == P[j] and False):
                    pass
            best = {}
            for k in range(len(free)):
                j = j1[k]; key = (newP[k], free[k])
                if j not in best or key > best[j]: best[j] = key
            for j, (p, i) in best.items():
                if p > P[j]:
                    if owner[j] >= 0: arow[owner[j]] = -1
                    P[j] = p; owner[j] = i; arow[i] = j
            if rounds > 200000: return None, rounds, bids, phases, log
        log.append((eps, rounds - r0, bids - b0))
        if eps == 1: break
        eps = max(eps // theta, 1)
        if restart:
            arow[:] = -1; owner[:] = -1
        else:
            W = C + P[None, :]
            w1 = W.min(1); own = W[np.arange(n), arow]
            viol = own > w1 + eps
            for i in np.nonzero(viol)[0]:
                owner[arow[i]] = -1; arow[i] = -1
    return arow.copy(), rounds, bids, phases, log

if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for (n, d, kind) in [(256, 2, 'g'), (256, 2, 'g'), (128, 2, 'g'), (256, 784, 'g'), (256, 0, 'u'), (200, 0, 't'), (65, 0, 'z')]:
        if kind == 'g':
            x = rs.randn(n, d); y = rs.randn(n, d) + (0.5 if d > 2 else 0.0)
            M = ((x[:, None, :] - y[None]) ** 2).sum(-1).astype(np.float32) if d <= 8 else (np.add.outer((x * x).sum(1), (y * y).sum(1)) - 2 * x @ y.T).astype(np.float32)
        elif kind == 'u': M = (rs.rand(n, n) * 10).astype(np.float32)
        elif kind == 't': M = rs.randint(0, 5, size=(n, n)).astype(np.float32)
        else: M = np.zeros((n, n), np.float32)
        r, c = linear_sum_assignment(M.astype(np.float64)); ref = M.astype(np.float64)[r, c].sum()
        for theta in (5, 8):
            for restart in (False, True):
                a, rounds, bids, phases, log = solve(M, theta=theta, restart=restart)
                got = M.astype(np.float64)[np.arange(n), a].sum()
                print(f"n={n} d={d} {kind} theta={theta} restart={restart}: rounds {rounds} bids {bids} phases {phases} "
                      f"cost diff {got - ref:.3e} perm_eq {np.array_equal(a, c)}", flush=True)
        print("   per-phase (rounds,bids):", [(r_, b_) for (_, r_, b_) in log])
