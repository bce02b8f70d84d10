# Forest Bellman-Ford ON CANDIDATE LISTS, entry-parallel rounds (every dirty column of a round is relaxed,
# no batch cap: one chip-wide launch per round), all the way down to 0 free rows: a model of replacing both
# the dense relax rounds and the one-workgroup list solver by list rounds.  Counts rounds / entries / dense
# fallbacks per phase; validated against SciPy's optimum.
import numpy as np, sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from proto import auction_phase
from proto4 import build_lists
from proto5 import bench_batch, cost32
from proto7 import col_reduce
from scipy.optimize import linear_sum_assignment as lsa


def forest_phase(C, p, a, owner, cols, T, radius_rule="max"):
    n = C.shape[0]
    roots = np.where(a < 0)[0]; freec = owner < 0
    ar = np.arange(n)
    u = np.where(a >= 0, C[ar, np.maximum(a, 0)] + p[np.maximum(a, 0)], 0.0)
    for r in roots: u[r] = (C[r] + p).min()
    d = np.full(n, np.inf); rnd = np.full(n, -1)
    # round 0: roots
    rounds = 0; entries = 0; dense = 0; sizes = []
    def apply(rows, base, dense_rows=False):
        nonlocal d
        if dense_rows:
            cand = base[:, None] + np.maximum(C[rows] + p[None, :] - u[rows][:, None], 0.0)
            kk = np.broadcast_to(ar[None, :], cand.shape)
        else:
            kk = cols[rows]
            cand = base[:, None] + np.maximum(C[rows[:, None], kk] + p[kk] - u[rows][:, None], 0.0)
        nd = d.copy()
        np.minimum.at(nd, kk.ravel(), cand.ravel())
        return nd
    def radius(dd):
        f = dd[freec]
        if len(roots) == 1 or radius_rule == "min": return f.min()
        return f.max()       # inf until every free column is reached
    nd = apply(roots, np.zeros(len(roots)))
    D = np.inf
    imp = (nd < d) & (nd < D); d = nd; rnd[imp] = rounds
    dirty = imp & (owner >= 0)
    rounds += 1; entries += len(roots); sizes.append(len(roots))
    dense_done = np.full(n, np.inf); root_dense = np.zeros(n, bool)
    while True:
        while True:
            D = radius(d)
            S = np.where(dirty & (d < D))[0]
            dirty[:] = False
            if len(S) == 0: break
            nd = apply(owner[S], d[S])
            imp = (nd < d) & (nd < D); d = np.where(imp, nd, d); rnd[imp] = rounds
            dirty = imp & (owner >= 0)
            rounds += 1; entries += len(S); sizes.append(len(S))
        D = radius(d)
        # a-posteriori test
        tree = np.where((owner >= 0) & (d < D) & (d < dense_done))[0]
        bad = tree[~(d[tree] + (T[owner[tree]] - u[owner[tree]]) >= D)]
        badr = roots[(~root_dense[roots]) & ~(T[roots] - u[roots] >= D)]
        if len(bad) == 0 and len(badr) == 0: break
        dense += len(bad) + len(badr)
        rows = np.concatenate([owner[bad], badr]); base = np.concatenate([d[bad], np.zeros(len(badr))])
        dense_done[bad] = d[bad]; root_dense[badr] = True
        nd = apply(rows, base, dense_rows=True)
        imp = (nd < d) & (nd < D); d = np.where(imp, nd, d); rnd[imp] = rounds
        dirty = imp & (owner >= 0)
        rounds += 1
    # recover predecessors (one more round in the kernel): tight edges
    D = radius(d)
    pred = np.full(n, -1)
    tr = np.where((owner >= 0) & (d <= D))[0]
    rows = np.concatenate([roots, owner[tr]]); base = np.concatenate([np.zeros(len(roots)), d[tr]])
    for i, b in zip(rows, base):
        if root_dense[i] or (a[i] >= 0 and dense_done[a[i]] < np.inf): ks = ar
        else: ks = cols[i]
        cand = b + np.maximum(C[i, ks] + p[ks] - u[i], 0.0)
        rj = rnd[a[i]] if a[i] >= 0 else -1
        hit = ks[(cand == d[ks]) & (d[ks] <= D) & ((b < d[ks]) | (rj < rnd[ks]))]
        upd = hit[(pred[hit] < 0) | (pred[hit] > i)]
        pred[upd] = i
    # accept one path per tree
    def true_root(k):
        i = pred[k]; g = 0
        assert i >= 0, ("nopred free", k, d[k], D, rnd[k])
        while a[i] >= 0:
            j = a[i]
            assert pred[j] >= 0, ("nopred", j, d[j], D, rnd[j], owner[j])
            i = pred[j]; g += 1
            assert g <= n, "cycle"
        return i
    best = {}
    for k in np.where(freec & np.isfinite(d) & (d <= D))[0]:
        r = true_root(k)
        if r not in best or d[k] < d[best[r]]: best[r] = k
    Dacc = max(d[k] for k in best.values())
    inT = d < Dacc; p[inT] += Dacc - d[inT]
    for r, k in best.items():
        j = k
        while True:
            i = pred[j]; owner[j] = i; jp = a[i]; a[i] = j
            if i == r: break
            j = jp
    return len(best), rounds, entries, dense, max(sizes)


def solve(M, K=64, verbose=True, radius_rule="max"):
    C = M.astype(np.float64); n = C.shape[0]; Cr = C.max() - C.min()
    u = C.min(1); p = -(C - u[:, None]).min(0)
    a = np.full(n, -1); owner = np.full(n, -1); eps = Cr * 8e-3; stats = []
    while eps >= Cr * 1e-6:
        a[:] = -1; owner[:] = -1
        auction_phase(C, p, a, owner, eps, 100000, int(0.02 * n), stats); eps /= 5
    a[:] = -1; owner[:] = -1; st = []
    auction_phase(C, p, a, owner, 0.0, 15, 0, st)
    col_reduce(C, p, owner)
    F0 = int((a < 0).sum())
    cols, T = build_lists(C, p, min(K, n - 1))
    ph = []
    while (a < 0).sum() > 0:
        ph.append(forest_phase(C, p, a, owner, cols, T, radius_rule))
    cost = C[np.arange(n), a].sum(); ref = C[lsa(C)].sum()
    ok = abs(cost - ref) <= 1e-9 * max(1, abs(ref)) and len(set(a)) == n
    uu = C[np.arange(n), a] + p[a]; slack = (C + p[None, :] - uu[:, None]).min()
    if verbose:
        print(f"   n={n} free after ARR {F0}; phases (augmented, rounds, entries, dense, widest) {ph}")
        print(f"      total rounds {sum(x[1] for x in ph)} (+{len(ph)} recover) entries {sum(x[2] for x in ph)} optimal {ok} minslack/Cr {slack/Cr:.1e}", flush=True)
    return ok


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    allok = True
    print("bench-like d=784")
    for kb in range(2):
        x0, x1 = bench_batch(n, 784, 1000, kb); t = time.time(); allok &= solve(cost32(x0, x1)); print("     t", time.time() - t)
    if len(sys.argv) > 2:
        print("uniform random")
        for nn in (300, 1000): allok &= solve(rng.random((nn, nn)).astype(np.float32))
        print("geometric d=2 (deep paths)")
        for nn in (256, 700):
            x = rng.standard_normal((nn, 2)); y = rng.standard_normal((nn, 2)) + 0.5
            allok &= solve(((x[:, None, :] - y[None]) ** 2).sum(-1).astype(np.float32))
        print("heavy ties (integer costs 0..9)")
        allok &= solve(rng.integers(0, 10, (400, 400)).astype(np.float32))
    print("ALL OPTIMAL" if allok else "MISMATCH")
