import sys, numpy as np
from scipy.optimize import linear_sum_assignment

def solve(M, theta=5, eps0_frac=2.0**-7, stop_frac=0.0, scale_bits=43):
    n = M.shape[0]
    cmin, cmax = float(M.min()), float(M.max())
    rng = cmax - cmin
    if rng == 0: rng = 1.0
    S = (n + 1) * 2.0**scale_bits / rng
    C = np.rint((M.astype(np.float64) - cmin) * S).astype(np.int64)
    P = np.zeros(n, np.int64); owner = -np.ones(n, np.int64); arow = -np.ones(n, np.int64)
    eps = max(int(C.max() * eps0_frac), 1)
    rounds = bids = phases = 0; log = []
    ar = np.arange(n)
    while True:
        phases += 1; r0 = rounds; b0 = bids
        stop = 0 if eps == 1 else int(stop_frac * n)
        while True:
            free = np.nonzero(arow < 0)[0]
            if len(free) <= stop: break
            rounds += 1; bids += len(free)
            W = C[free] + P[None, :]
            j1 = W.argmin(1); w1 = W[np.arange(len(free)), j1]
            W[np.arange(len(free)), j1] = np.iinfo(np.int64).max
            w2 = W.min(1)
            newP = P[j1] + (w2 - w1) + eps
            best = {}
            for k in range(len(free)):
                j = j1[k]; key = (newP[k], free[k])
                if j not in best or key > best[j]: best[j] = key
            for j, (p, i) in best.items():
                if p > P[j]:
                    if owner[j] >= 0: arow[owner[j]] = -1
                    P[j] = p; owner[j] = i; arow[i] = j
            if rounds > 100000: return None, rounds, bids, phases, log
        log.append((rounds - r0, bids - b0))
        if eps == 1: break
        eps = max(eps // theta, 1)
        W = C + P[None, :]
        w1 = W.min(1)
        asg = arow >= 0
        own = np.where(asg, W[ar, np.maximum(arow, 0)], 0)
        viol = asg & (own > w1 + eps)
        for i in np.nonzero(viol)[0]:
            owner[arow[i]] = -1; arow[i] = -1
    return arow.copy(), rounds, bids, phases, log

rs = np.random.RandomState(1)
cases = []
for (n, d, kind) in [(256, 2, 'g'), (256, 2, 'g'), (256, 2, 'g'), (128, 2, 'g'), (256, 784, 'g'), (256, 0, 'u'), (200, 0, 't')]:
    if kind == 'g':
        x = rs.randn(n, d); y = rs.randn(n, d) + (0.5 if d > 2 else 0.0)
        M = ((x[:, None, :] - y[None]) ** 2).sum(-1).astype(np.float32) if d <= 8 else (np.add.outer((x * x).sum(1), (y * y).sum(1)) - 2 * x @ y.T).astype(np.float32)
    elif kind == 'u': M = (rs.rand(n, n) * 10).astype(np.float32)
    else: M = rs.randint(0, 5, size=(n, n)).astype(np.float32)
    cases.append((n, d, kind, M))
for theta in (4, 5, 8, 16):
    for stop in (0.0, 0.02, 0.05):
        out = []
        for (n, d, kind, M) in cases:
            r, c = linear_sum_assignment(M.astype(np.float64)); ref = M.astype(np.float64)[r, c].sum()
            a, rounds, bids, phases, log = solve(M, theta=theta, stop_frac=stop)
            ok = a is not None and abs(M.astype(np.float64)[np.arange(n), a].sum() - ref) < 1e-9 * max(1, abs(ref))
            out.append(f"{rounds}/{bids}{'' if ok else '!'}")
        print(f"theta={theta} stop={stop}: " + "  ".join(out), flush=True)
