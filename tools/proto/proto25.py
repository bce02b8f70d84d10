"""Prototype: the transportation solver (proto20 / transport.hip) with MULTI-SOURCE phases — one shortest-path forest
grown from ALL rows with supply left (a column belongs to the tree that labelled it; every support row of a scanned
column joins that tree), every tree that reaches a column with demand left takes its nearest one, the duals move by
D = the largest accepted label, all accepted paths (vertex disjoint: the trees are) push their bottlenecks.  A phase
relaxes every row at most once.  Counts phases / row relaxations against the one-search-per-augmentation form."""
import math, sys, time
import numpy as np
from proto20 import clouds


def solve(M):
    M = M.astype(np.float64)
    B0, B1 = M.shape
    g = math.gcd(B0, B1); p, q = B1 // g, B0 // g
    rs = np.full(B0, p, dtype=np.int64); rd = np.full(B1, q, dtype=np.int64)
    u = M.min(1).copy(); v = np.zeros(B1)
    F = [dict() for _ in range(B1)]          # column -> {row: units}
    am = M.argmin(1)
    for i in range(B0):
        j = am[i]; d = min(rs[i], rd[j])
        if d > 0: F[j][i] = d; rs[i] -= d; rd[j] -= d
    st = dict(phases=0, relax=0, augment=0)
    while rs.sum() > 0:
        st["phases"] += 1
        roots = np.where(rs > 0)[0]
        dist = np.full(B1, np.inf); pred = np.full(B1, -1); tree_c = np.full(B1, -1)
        dr = np.full(B0, np.inf); par = np.full(B0, -1); tree_r = np.full(B0, -1)
        for t, r in enumerate(roots):
            dr[r] = 0.0; tree_r[r] = t
            nd = M[r] - u[r] - v
            b = nd < dist
            dist[b] = nd[b]; pred[b] = r; tree_c[b] = t
            st["relax"] += 1
        scanned = np.zeros(B1, bool)
        sink = {}                      # tree -> (label, column)
        radius = np.inf
        while True:
            dm = np.where(scanned, np.inf, dist)
            j = int(dm.argmin()); D = dm[j]
            if not np.isfinite(D) or D > radius: break
            scanned[j] = True
            t = tree_c[j]
            if rd[j] > 0 and t not in sink:
                sink[t] = (D, j)
                if len(sink) == len(roots): radius = max(x[0] for x in sink.values())
            # (every label below the radius has to become final: keep expanding, also through columns with demand left)
            for i in F[j]:
                if F[j][i] > 0 and tree_r[i] < 0:
                    tree_r[i] = t; dr[i] = D; par[i] = j
                    nd = D + M[i] - u[i] - v
                    b = (nd < dist) & ~scanned
                    dist[b] = nd[b]; pred[b] = i; tree_c[b] = t
                    st["relax"] += 1
        if not sink: raise RuntimeError("no sink")
        Dm = max(x[0] for x in sink.values())
        # duals: everything labelled below Dm
        rows_in = np.where(dr < Dm)[0]; u[rows_in] += Dm - dr[rows_in]
        cols_in = scanned & (dist < Dm); v[cols_in] += dist[cols_in] - Dm
        for t, (lab, j) in sink.items():
            r = roots[t]
            path = []; jj = j
            while True:
                i = int(pred[jj]); path.append((i, jj))
                if par[i] < 0: break
                jj = par[i]
            if path[-1][0] != r: continue      # (a path that left its tree: skip, the next phase takes it)
            d = min(rs[r], rd[j])
            for (i, jc) in path:
                if par[i] >= 0: d = min(d, F[par[i]].get(i, 0))
            if d <= 0: continue
            for (i, jc) in path:
                F[jc][i] = F[jc].get(i, 0) + d
                if par[i] >= 0:
                    F[par[i]][i] -= d
            rs[r] -= d; rd[j] -= d; st["augment"] += 1
    cost = sum(f * M[i, j] for j in range(B1) for i, f in F[j].items()) / (B0 * p)
    rc = M - u[:, None] - v[None, :]
    assert rc.min() > -1e-9, rc.min()
    return cost, st


if __name__ == "__main__":
    B0, B1, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    M = clouds(B0, B1, d, 0)
    t = time.time(); c, st = solve(M); t = time.time() - t
    print(B0, B1, d, "cost", c, st, f"{t:.1f}s")
    import proto20
    c2, st2 = proto20.solve(M)
    print("single-search form: cost", c2, st2, "diff", c - c2)
