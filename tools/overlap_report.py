"""How well do the couplings in flight overlap?  Reads a rocprofv3 --kernel-trace directory of a pipelined bench run and
prints, per queue, the busy time, and over all queues the time with 0 / 1 / 2 / 3+ kernels resident, plus the same
restricted to the intervals in which a one-workgroup list solver (asg_solve) is running.
    python tools/overlap_report.py <dir> [t_skip_fraction]
Measurement infrastructure."""
import collections, csv, glob, os, sys


def main(d, skip=0.3):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rd = csv.DictReader(fh)
            cols = rd.fieldnames
            for r in rd:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40],
                             r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
    print("columns:", cols)
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo = t0 + skip * (t1 - t0)                # drop the warm-up part
    rows = [r for r in rows if r[0] >= lo]
    span = max(r[1] for r in rows) - rows[0][0]
    print(f"{len(rows)} kernels over {span/1e6:.2f} ms")
    perq = collections.defaultdict(float)
    for s, e, k, q, st in rows:
        perq[(q, st)] += e - s
    for q, v in sorted(perq.items()):
        print(f"  queue/stream {q}: busy {v/1e6:.2f} ms = {100*v/span:.1f} % of the span")
    ev = []
    for s, e, k, q, st in rows:
        ev.append((s, 1, k == "asg_solve")); ev.append((e, -1, k == "asg_solve"))
    ev.sort()
    depth = 0; solve = 0; last = ev[0][0]
    hist = collections.defaultdict(float); hist_s = collections.defaultdict(float); nsolve = collections.defaultdict(float)
    for t, dlt, is_solve in ev:
        hist[min(depth, 4)] += t - last
        if solve:
            hist_s[min(depth, 4)] += t - last
        nsolve[min(solve, 4)] += t - last
        last = t; depth += dlt
        if is_solve:
            solve += dlt
    print("kernels resident:   " + "  ".join(f"{k}: {100*v/span:.1f} %" for k, v in sorted(hist.items())))
    tot_s = sum(hist_s.values()) or 1
    print("while a list solver runs: " + "  ".join(f"{k}: {100*v/tot_s:.1f} %" for k, v in sorted(hist_s.items())))
    print("list solvers resident:  " + "  ".join(f"{k}: {100*v/span:.1f} %" for k, v in sorted(nsolve.items())))
    # gaps between consecutive kernels of one stream, by size class: many small gaps = launch boundaries, few large = host
    bystream = collections.defaultdict(list)
    for s, e, k, q, st in rows:
        bystream[(q, st)].append((s, e, k))
    edges = [3e3, 10e3, 50e3, 200e3, 1e6, 1e12]
    for q, lst in sorted(bystream.items()):
        lst.sort()
        cls = [[0, 0.0] for _ in edges]
        after = collections.defaultdict(float)
        for (s0, e0, k0), (s1, e1, k1) in zip(lst, lst[1:]):
            g = max(0, s1 - e0)
            for c, ed in enumerate(edges):
                if g < ed:
                    cls[c][0] += 1; cls[c][1] += g; break
            if g >= 50e3:
                after[f"{k0} -> {k1}"] += g
        print(f"  stream {q} gaps: " + "  ".join(f"<{ed/1e3:g}us: {n} ({t/1e6:.1f} ms)" for ed, (n, t) in zip(edges, cls)))
        for k, v in sorted(after.items(), key=lambda kv: -kv[1])[:6]:
            print(f"      large gaps after/before {k}: {v/1e6:.1f} ms")
    agg = collections.defaultdict(list)
    for s, e, k, q, st in rows:
        agg[k].append(e - s)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print(f"  {k:40s} calls {len(v):6d}  total {sum(v)/1e6:8.2f} ms  avg {sum(v)/len(v)/1e3:8.2f} us")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
