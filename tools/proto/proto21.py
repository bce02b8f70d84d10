"""Prototype: how many bids of the epsilon-scaling auction (C3) could be decided from a 64-entry candidate list?

Emulates the device auction's schedule (JV column-reduction start, eps0 = 8e-3 of the cost range, theta = 5, eps_last = 1e-6,
all rows unassigned again at every phase start, Jacobi rounds, one winner per object) on the CPU and classifies every
bid against a list of the K smallest (c + p) of its row — built once after the start prices, or rebuilt at every phase
start (when every row scans its whole row anyway).  T = smallest (c + p) outside the list at build time; prices only
rise, so T stays a lower bound of everything outside:
    exact   : second best inside the list <= T      -> the list decides the bid exactly
    partial : best inside <= T < second best inside -> best is right, increment can use min(second, T) (a smaller, still valid bid)
    fail    : best inside > T                       -> full row scan
"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import cfm_oracle as oracle


def run(C, K=64, rebuild="phase", theta=5.0, eps0_frac=8e-3, eps_last_frac=1e-6, stop_frac=0.02, round_cap=4000, use_partial=True):
    n = C.shape[0]
    C = C.astype(np.float64)
    rng = C.max() - C.min()
    u = C.min(1)
    p = (u[:, None] - C).max(0)            # JV column reduction: p_j = max_i (u_i - c_ij) <= 0
    eps = eps0_frac * rng
    owner = np.full(n, -1); a = np.full(n, -1)
    stats = dict(rounds=0, bids=0, exact=0, partial=0, fail=0, full_scans=0, phase_starts=0)

    def build(rows):
        v = C[rows] + p
        idx = np.argpartition(v, K, axis=1)
        mem = idx[:, :K]
        T = np.take_along_axis(v, idx[:, K:], 1).min(1)
        return mem, T
    mem, T = build(np.arange(n))
    stats["full_scans"] += n
    phase = 0
    while True:
        # phase start: all rows unassigned
        owner[:] = -1; a[:] = -1
        stop = int(stop_frac * n)
        rnd = 0
        first = True
        while True:
            free = np.where(a < 0)[0]
            cnt = len(free)
            if cnt == 0: break
            if not first and (cnt <= stop or rnd >= round_cap): break
            if first and rebuild == "phase" and phase > 0:
                mem, T = build(np.arange(n)); stats["phase_starts"] += 1
            stats["rounds"] += 1; stats["bids"] += cnt
            # list evaluation
            lv = np.take_along_axis(C[free], mem[free], 1) + p[mem[free]]
            o = np.argsort(lv, axis=1)[:, :2]
            b1 = np.take_along_axis(lv, o[:, :1], 1)[:, 0]; s1 = np.take_along_axis(lv, o[:, 1:2], 1)[:, 0]
            j1 = np.take_along_axis(mem[free], o[:, :1], 1)[:, 0]
            ex = s1 <= T[free]
            pa = (~ex) & (b1 <= T[free])
            fl = ~(ex | pa)
            if first and phase > 0 and rebuild == "phase":
                pass
            stats["exact"] += int(ex.sum()); stats["partial"] += int(pa.sum()); stats["fail"] += int(fl.sum())
            # true values for the rows that need them
            need = fl if use_partial else ~ex
            bj = j1.copy(); bb = b1.copy(); ss = np.where(ex, s1, np.minimum(s1, T[free]))
            if need.any():
                rows = free[need]
                v = C[rows] + p
                oo = np.argpartition(v, 1, axis=1)[:, :2]
                v2 = np.take_along_axis(v, oo, 1)
                sw = v2[:, 0] > v2[:, 1]
                jb = np.where(sw, oo[:, 1], oo[:, 0]); vb = v2.min(1); vs = v2.max(1)
                bj[need] = jb; bb[need] = vb; ss[need] = vs
                stats["full_scans"] += int(need.sum())
            if first and rebuild == "phase" and phase > 0:
                stats["full_scans"] += 0      # the rebuild scan is counted in phase_starts
            bid = p[bj] + (ss - bb) + eps
            # one winner per object: highest bid, ties -> higher row
            order = np.lexsort((free, bid))
            win = {}
            for k in order: win[bj[k]] = k
            for j, k in win.items():
                i = free[k]
                if bid[k] > p[j] or owner[j] < 0:
                    if owner[j] >= 0: a[owner[j]] = -1
                    owner[j] = i; a[i] = j; p[j] = max(p[j], bid[k])
            rnd += 1; first = False
        e2 = eps / theta
        if e2 < eps_last_frac * rng: break
        eps = e2; phase += 1
    stats["free_at_end"] = int((a < 0).sum())
    return stats


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    x0, x1 = oracle.config_inputs("C3", B=B)
    C = oracle.ref_cost_f32(x0, x1)
    for K in (64, 32):
        for rebuild in ("once", "phase"):
            t = time.time(); st = run(C, K=K, rebuild=rebuild); t = time.time() - t
            print(f"K={K} rebuild={rebuild}: {st}  ({t:.0f}s)")
