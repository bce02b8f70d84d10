"""Prototype: exact transportation between uniform marginals of different sizes WITHOUT the lcm expansion.

Integer form: every row supplies p = B1/g units, every column demands q = B0/g units (g = gcd), unit mass 1/(B0*p).
Successive shortest paths on the B0 x B1 matrix with column labels (dist per column, every support row of a scanned
column joins the tree at the column's distance), duals kept feasible, flow only on tight entries.  Counts the row scans
(the unit of work of a one-workgroup device kernel) per variant.
"""
import math, sys, time
import numpy as np


def clouds(B0, B1, d, seed):
    r = np.random.default_rng(seed)
    x0 = r.standard_normal((B0, d)).astype(np.float32)
    x1 = (r.standard_normal((B1, d)) * 0.7 + 0.5).astype(np.float32)
    M = ((x0[:, None, :] - x1[None, :, :]) ** 2).sum(-1) if B0 * B1 * d < 5e7 else None
    if M is None:
        M = (x0 ** 2).sum(1)[:, None] + (x1 ** 2).sum(1)[None, :] - 2.0 * x0 @ x1.T
        M = np.maximum(M, 0)
    return M.astype(np.float32)


def solve(M, init="greedy", order="index", verbose=False):
    M = M.astype(np.float64)
    B0, B1 = M.shape
    g = math.gcd(B0, B1)
    p, q = B1 // g, B0 // g
    rs = np.full(B0, p, dtype=np.int64)
    rd = np.full(B1, q, dtype=np.int64)
    u = M.min(1).copy()
    v = np.zeros(B1)
    F = {}                      # (i, j) -> units
    colrows = [dict() for _ in range(B1)]     # j -> {i: units}
    stats = dict(searches=0, scans=0, augment_edges=0, init_units=0)

    def push(i, j, dlt):
        colrows[j][i] = colrows[j].get(i, 0) + dlt
        if colrows[j][i] == 0:
            del colrows[j][i]

    if init == "greedy":
        am = M.argmin(1)
        for i in range(B0):
            j = am[i]
            dlt = min(rs[i], rd[j])
            if dlt > 0:
                push(i, j, dlt); rs[i] -= dlt; rd[j] -= dlt; stats["init_units"] += dlt
    rows = list(range(B0))
    for r in rows:
        while rs[r] > 0:
            stats["searches"] += 1
            dist = M[r] - u[r] - v
            pred = np.full(B1, r)
            scanned = np.zeros(B1, bool)
            dr = {r: 0.0}
            parent = {r: -1}
            stats["scans"] += 1
            while True:
                dm = np.where(scanned, np.inf, dist)
                j = int(dm.argmin()); D = dm[j]
                if rd[j] > 0:
                    break
                scanned[j] = True
                for i in colrows[j]:
                    if i not in dr:
                        dr[i] = D; parent[i] = j
                        nd = D + M[i] - u[i] - v
                        better = (nd < dist) & ~scanned
                        dist = np.where(better, nd, dist); pred = np.where(better, i, pred)
                        stats["scans"] += 1
            # bottleneck
            dlt = min(rs[r], rd[j])
            jj = j; path = []
            while True:
                i = int(pred[jj]); path.append((i, jj))
                if parent[i] < 0:
                    break
                dlt = min(dlt, colrows[parent[i]][i]); jj = parent[i]
            for i, D_i in dr.items():
                u[i] += D - D_i
            v[scanned] += dist[scanned] - D
            jj = j
            for (i, jc) in path:
                push(i, jc, dlt)
                if parent[i] >= 0:
                    push(i, parent[i], -dlt)
            stats["augment_edges"] += len(path)
            rs[r] -= dlt; rd[j] -= dlt
    cost = 0.0; nsup = 0
    for j in range(B1):
        for i, f in colrows[j].items():
            cost += f * M[i, j]; nsup += 1
    cost /= (B0 * p)
    # certificate
    rc = M - u[:, None] - v[None, :]
    assert rc.min() > -1e-9, rc.min()
    for j in range(B1):
        assert sum(colrows[j].values()) == q
        for i in colrows[j]:
            assert abs(rc[i, j]) < 1e-9
    stats["support"] = nsup
    return cost, stats


if __name__ == "__main__":
    B0, B1, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    M = clouds(B0, B1, d, 0)
    t = time.time(); c, st = solve(M); t = time.time() - t
    print(B0, B1, d, "cost", c, st, f"{t:.1f}s", "scans/search", st["scans"] / max(1, st["searches"]))
    if B0 * B1 <= 130 * 130:
        from scipy.optimize import linprog
        import scipy.sparse as sp
        A = sp.vstack([sp.kron(sp.eye(B0), np.ones((1, B1))), sp.kron(np.ones((1, B0)), sp.eye(B1))]).tocsr()
        b = np.concatenate([np.full(B0, 1.0 / B0), np.full(B1, 1.0 / B1)])
        r = linprog(M.astype(np.float64).ravel(), A_eq=A, b_eq=b, bounds=(0, None), method="highs")
        print("linprog", r.fun, "diff", c - r.fun)
