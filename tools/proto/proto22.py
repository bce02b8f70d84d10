"""Prototype: candidate lists that cost nothing to build — the per-lane best (or best two) of a full row scan.

A bidding wave's full scan keeps, per lane, the top-2 (or top-3) of c + p over the lane's 64 columns (lane l owns columns
256 k + 4 l + e).  List = every lane's best m entries (m = 1: 64 entries, m = 2: 128), T = the smallest lane-(m+1)-th value:
a lower bound of everything outside the list, valid while prices only rise.  Every full scan refreshes the row's list.
Counts list-decided bids vs full scans over the whole epsilon-scaling auction at C3 (schedule as in proto21)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import cfm_oracle as oracle


def lane_of(n):
    j = np.arange(n)
    return (j % 256) // 4


def run(C, m=1, theta=5.0, eps0_frac=8e-3, eps_last_frac=1e-6, stop_frac=0.02, use_partial=True, refresh_phase=False):
    n = C.shape[0]
    C = C.astype(np.float64)
    rng = C.max() - C.min()
    u = C.min(1)
    p = (u[:, None] - C).max(0)
    eps = eps0_frac * rng
    ln = lane_of(n)
    perm = np.argsort(ln, kind="stable")           # columns grouped by lane: [64 lanes, n/64]
    per = n // 64
    Cl = C[:, perm]
    owner = np.full(n, -1); a = np.full(n, -1)
    mem = np.zeros((n, 64 * m), dtype=np.int64); T = np.full(n, -np.inf)
    st = dict(rounds=0, bids=0, exact=0, partial=0, fail=0, full_scans=0)

    def scan(rows):
        """full scan of `rows`: true top-2 and the refreshed list"""
        v = (Cl[rows] + p[perm]).reshape(len(rows), 64, per)
        o = np.argsort(v, axis=2)[:, :, :m + 1]
        vs = np.take_along_axis(v, o, 2)
        cols = perm[(np.arange(64)[None, :, None] * per + o)]
        mem[rows] = cols[:, :, :m].reshape(len(rows), -1)
        T[rows] = vs[:, :, m].min(1)
        flat = vs[:, :, :2].reshape(len(rows), -1); fcol = cols[:, :, :2].reshape(len(rows), -1)
        oo = np.argsort(flat, axis=1)[:, :2]
        b = np.take_along_axis(flat, oo[:, :1], 1)[:, 0]; s = np.take_along_axis(flat, oo[:, 1:2], 1)[:, 0]
        j = np.take_along_axis(fcol, oo[:, :1], 1)[:, 0]
        st["full_scans"] += len(rows)
        return j, b, s
    phase = 0
    while True:
        owner[:] = -1; a[:] = -1
        stop = int(stop_frac * n); first = True
        while True:
            free = np.where(a < 0)[0]; cnt = len(free)
            if cnt == 0 or (not first and cnt <= stop): break
            st["rounds"] += 1; st["bids"] += cnt
            if (phase == 0 and first) or (refresh_phase and first):
                bj, bb, ss = scan(free)
            else:
                lv = np.take_along_axis(C[free], mem[free], 1) + p[mem[free]]
                o = np.argsort(lv, axis=1)[:, :2]
                bb = np.take_along_axis(lv, o[:, :1], 1)[:, 0]; s1 = np.take_along_axis(lv, o[:, 1:2], 1)[:, 0]
                bj = np.take_along_axis(mem[free], o[:, :1], 1)[:, 0]
                ex = s1 <= T[free]; pa = (~ex) & (bb <= T[free]); fl = ~(ex | pa)
                st["exact"] += int(ex.sum()); st["partial"] += int(pa.sum()); st["fail"] += int(fl.sum())
                ss = np.where(ex, s1, np.minimum(s1, T[free]))
                need = fl if use_partial else ~ex
                if need.any():
                    j2, b2, s2 = scan(free[need])
                    bj[need] = j2; bb[need] = b2; ss[need] = s2
            bid = p[bj] + (ss - bb) + eps
            order = np.lexsort((free, bid))
            win = {}
            for k in order: win[bj[k]] = k
            for j, k in win.items():
                i = free[k]
                if bid[k] > p[j] or owner[j] < 0:
                    if owner[j] >= 0: a[owner[j]] = -1
                    owner[j] = i; a[i] = j; p[j] = max(p[j], bid[k])
            first = False
        e2 = eps / theta
        if e2 < eps_last_frac * rng: break
        eps = e2; phase += 1
    st["free_at_end"] = int((a < 0).sum())
    return st


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    x0, x1 = oracle.config_inputs("C3", B=B)
    C = oracle.ref_cost_f32(x0, x1)
    for m in (1, 2):
        for up in (True, False):
            t = time.time(); st = run(C, m=m, use_partial=up); t = time.time() - t
            print(f"m={m} partial_bids={up}: {st} ({t:.0f}s)")
