"""Discrete-event emulation of two schedules of the one-workgroup auction (16 waves; row i belongs to wave i % 16;
a bid costs one time unit of its wave): (a) Jacobi rounds with a barrier per round (time of a round = the largest
number of unmatched rows of any wave), (b) barrier-free: every wave keeps bidding for its own unmatched rows with
the prices as they are (asynchronous auction); a phase ends when <= 2 % of the rows are unmatched.
Prints total bids and the critical path (time units) of the epsilon phases down to eps_last."""
import sys, numpy as np

def instance(kind, n, rs):
    if kind == "g2":
        x = rs.randn(n, 2); y = rs.randn(n, 2)
        return ((x[:, None, :] - y[None]) ** 2).sum(-1)
    if kind == "g784":
        x = rs.randn(n, 784); y = rs.randn(n, 784) + 0.5
        return np.add.outer((x * x).sum(1), (y * y).sum(1)) - 2 * x @ y.T
    return rs.rand(n, n) * 10

def colred(C):
    u = C.min(1)
    return (u[:, None] - C).max(0)

rounds_box = [0]
def run(C, mode, theta=5.0, eps0=8e-3, epsl=1e-6, stop_frac=0.02, NW=16):
    n = C.shape[0]; rng = C.max() - C.min()
    P = colred(C); P = P - P.min()
    eps = eps0 * rng; eps_last = epsl * rng; stop = int(stop_frac * n)
    bids = 0; timeu = 0; rounds_box[0] = 0
    while True:
        owner = -np.ones(n, int); arow = -np.ones(n, int)
        if mode == "jacobi":
            while True:
                free = np.nonzero(arow < 0)[0]
                if len(free) <= stop: break
                loads = np.bincount(free % NW, minlength=NW); timeu += loads.max(); bids += len(free)
                W = C[free] + P[None, :]
                j1 = W.argmin(1); w1 = W[np.arange(len(free)), j1]
                W[np.arange(len(free)), j1] = np.inf; w2 = W.min(1)
                newP = P[j1] + (w2 - w1) + eps
                best = {}
                for k in range(len(free)):
                    key = (newP[k], free[k])
                    if j1[k] not in best or key > best[j1[k]]: best[j1[k]] = key
                for j, (p, i) in best.items():
                    if p > P[j]:
                        if owner[j] >= 0: arow[owner[j]] = -1
                        P[j] = p; owner[j] = i; arow[i] = j
        elif mode.startswith("cap"):
            cap = int(mode[3:])
            while True:
                free = np.nonzero(arow < 0)[0]
                if len(free) <= stop: break
                # Jacobi round in which every wave bids for at most `cap` of its unmatched rows (snapshot prices)
                sel = []; cntw = np.zeros(NW, int)
                for i in free:
                    if cntw[i % NW] < cap: sel.append(i); cntw[i % NW] += 1
                sel = np.array(sel); timeu += cntw.max(); bids += len(sel); rounds_box[0] += 1
                W = C[sel] + P[None, :]
                j1 = W.argmin(1); w1 = W[np.arange(len(sel)), j1]
                W[np.arange(len(sel)), j1] = np.inf; w2 = W.min(1)
                newP = P[j1] + (w2 - w1) + eps
                best = {}
                for k in range(len(sel)):
                    key = (newP[k], sel[k])
                    if j1[k] not in best or key > best[j1[k]]: best[j1[k]] = key
                for j, (p, i) in best.items():
                    if p > P[j]:
                        if owner[j] >= 0: arow[owner[j]] = -1
                        P[j] = p; owner[j] = i; arow[i] = j
        else:
            while True:
                free = np.nonzero(arow < 0)[0]
                if len(free) <= stop: break
                timeu += 1
                # every wave with an unmatched row bids for its first one, waves in order, prices as they are
                done_w = set()
                for i in free:
                    w = i % NW
                    if w in done_w or arow[i] >= 0: continue
                    done_w.add(w); bids += 1
                    Wr = C[i] + P
                    j = int(Wr.argmin()); w1 = Wr[j]; Wr[j] = np.inf; w2 = Wr.min()
                    p = P[j] + (w2 - w1) + eps
                    if owner[j] >= 0: arow[owner[j]] = -1
                    P[j] = p; owner[j] = i; arow[i] = j
        e2 = eps / theta
        if e2 < eps_last: break
        eps = e2
    return bids, timeu

if __name__ == "__main__":
    rs = np.random.RandomState(0)
    for kind, n in (("g2", 256), ("g2", 256), ("g2", 128), ("g784", 256), ("u", 256)):
        C = instance(kind, n, rs)
        bj, tj = run(C, "jacobi"); ba, ta = run(C, "async")
        b1, t1 = run(C, "cap1"); r1 = rounds_box[0]; b2, t2 = run(C, "cap2"); r2 = rounds_box[0]
        print(f"{kind} n={n}: jacobi bids {bj} critical path {tj} | async bids {ba} critical path {ta} | "
              f"capped rounds: cap 1 {r1} rounds ({b1} bids), cap 2 {r2} rounds, path {t2} ({b2} bids)")
