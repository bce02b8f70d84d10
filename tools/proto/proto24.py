"""Prototype: epsilon-scaling auction for the TRANSPORTATION problem between uniform marginals of different sizes,
on the compact B0 x B1 problem (no lcm expansion).

Integer form: every row supplies p = B1/g units, every column takes q = B0/g (g = gcd(B0, B1)).
State per column j: its held unit groups (row, units, level) — at most q of them, every group has >= 1 unit — and the
price pi_j = the lowest held level once the column is full (its start price before).  A Jacobi round:
  rows     every row with free units finds its best column j1 and the second-best value w2 over the OTHER columns and
           bids ALL its free units for j1 at the level  b = w2 - c[i, j1] + eps   (value of j1 at that level = w2 + eps);
  columns  every column merges the bids addressed to it with what it holds, keeps the q units of highest level, sends
           the rest back to their rows, and raises its price to the lowest level kept if it is full.
Every held unit satisfies eps-complementary slackness against the column prices (its level is >= the price of its
column, and was its owner's indifference level + eps when it was bid; prices only rise), so when all units are placed
the plan's cost is within eps of the optimum (per unit of mass 1/L: within eps in total).  eps-scaling: theta = 5 from
8e-3 of the cost range down to eps_last; at a phase start every unit is free again, the prices stay.
Counts rounds / bids, checks the marginals and compares the cost with the LP optimum (HiGHS) where that is feasible."""
import math, sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def clouds(B0, B1, d, seed):
    r = np.random.default_rng(seed)
    x0 = r.standard_normal((B0, d)).astype(np.float32)
    x1 = (r.standard_normal((B1, d)) * 0.7 + 0.5).astype(np.float32)
    M = (x0 ** 2).sum(1)[:, None] + (x1 ** 2).sum(1)[None, :] - 2.0 * (x0 @ x1.T)
    return np.maximum(M, 0).astype(np.float32)


def auction(M, theta=5.0, eps0_frac=8e-3, eps_last_frac=1e-9, stop_frac=0.0, verbose=False):
    C = M.astype(np.float64)
    B0, B1 = C.shape
    g = math.gcd(B0, B1); p, q = B1 // g, B0 // g
    rng = C.max() - C.min()
    pi = np.zeros(B1)
    eps = eps0_frac * rng
    st = dict(rounds=0, bids=0, phases=0)
    while True:
        free = np.full(B0, p, dtype=np.int64)
        # held groups per column: arrays (row, units, level)
        hrow = [np.zeros(0, np.int64) for _ in range(B1)]
        hun = [np.zeros(0, np.int64) for _ in range(B1)]
        hlev = [np.zeros(0) for _ in range(B1)]
        st["phases"] += 1
        stop = int(stop_frac * B0 * p)
        while True:
            rows = np.where(free > 0)[0]
            if len(rows) == 0 or free.sum() <= stop and st["rounds"] > 0 and stop > 0:
                break
            st["rounds"] += 1; st["bids"] += len(rows)
            v = C[rows] + pi
            o = np.argpartition(v, 1, axis=1)[:, :2]
            v2 = np.take_along_axis(v, o, 1)
            sw = v2[:, 0] > v2[:, 1]
            j1 = np.where(sw, o[:, 1], o[:, 0]); w1 = v2.min(1); w2 = v2.max(1)
            lev = w2 - C[rows, j1] + eps                # >= pi[j1] + eps
            units = free[rows].copy(); free[rows] = 0
            order = np.argsort(j1, kind="stable")
            j1s = j1[order]; cut = np.flatnonzero(np.diff(j1s)) + 1
            for seg in np.split(order, cut):
                j = int(j1[seg[0]])
                r_ = np.concatenate([hrow[j], rows[seg]]); u_ = np.concatenate([hun[j], units[seg]]); l_ = np.concatenate([hlev[j], lev[seg]])
                k = np.argsort(-l_, kind="stable")
                r_, u_, l_ = r_[k], u_[k], l_[k]
                cum = np.cumsum(u_)
                keep = cum - u_ < q                     # groups that start inside the capacity
                over = np.where(keep, np.maximum(cum - q, 0), u_)      # units of each group beyond the capacity
                back = over > 0
                if back.any():
                    np.add.at(free, r_[back], over[back])
                u_ = u_ - over
                nz = u_ > 0
                hrow[j], hun[j], hlev[j] = r_[nz], u_[nz], l_[nz]
                if hun[j].sum() == q:
                    pi[j] = max(pi[j], hlev[j].min())
        e2 = eps / theta
        if eps <= eps_last_frac * rng * 1.0000001:
            break
        eps = max(e2, eps_last_frac * rng)
    # plan
    F = np.zeros((B0, B1), dtype=np.int64)
    for j in range(B1):
        np.add.at(F[:, j], hrow[j], hun[j])
    assert (F.sum(1) == p).all() and (F.sum(0) == q).all()
    cost = float((F * C).sum()) / (B0 * p)
    st["support"] = int((F > 0).sum()); st["eps_last_abs"] = eps
    return cost, F, pi, st


if __name__ == "__main__":
    B0, B1, d = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    el = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-9
    M = clouds(B0, B1, d, 0)
    t = time.time(); c, F, pi, st = auction(M, eps_last_frac=el); t = time.time() - t
    print(B0, B1, d, "cost", repr(c), st, f"{t:.1f}s")
    if B0 * B1 <= 300 * 300:
        from scipy.optimize import linprog
        import scipy.sparse as sp
        A = sp.vstack([sp.kron(sp.eye(B0), np.ones((1, B1))), sp.kron(np.ones((1, B0)), sp.eye(B1))]).tocsr()
        b = np.concatenate([np.full(B0, 1.0 / B0), np.full(B1, 1.0 / B1)])
        r = linprog(M.astype(np.float64).ravel(), A_eq=A, b_eq=b, bounds=(0, None), method="highs")
        print("linprog", r.fun, "gap", c - r.fun, "rel", (c - r.fun) / r.fun)
