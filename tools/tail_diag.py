"""Where do the slow regions of the pipelined loop come from?  Reads a rocprofv3 directory holding a kernel trace and
(optionally) a HIP API trace of a bench run and prints the longest kernel instances, per-kernel max / p99 / avg, the
longest HIP API calls, and the longest device-idle gaps with what ran before / after them.
    python tools/tail_diag.py <dir>
Measurement infrastructure."""
import collections, csv, glob, os, sys


def rows(d, pat):
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def main(d):
    ks = []
    for r in rows(d, "*kernel_trace.csv"):
        ks.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:48], r.get("Stream_Id", "?"),
                   r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
    ks.sort()
    if not ks:
        print("no kernel trace"); return
    t0 = ks[0][0]
    print(f"{len(ks)} kernels over {(ks[-1][1] - t0) / 1e6:.1f} ms")
    agg = collections.defaultdict(list)
    for s, e, k, st, wg, gr in ks:
        agg[k].append(e - s)
    print("per kernel: calls  avg_us  p99_us  max_us  total_ms")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:24]:
        v2 = sorted(v)
        print(f"  {k:48s} {len(v):7d} {sum(v) / len(v) / 1e3:9.1f} {v2[int(0.99 * (len(v2) - 1))] / 1e3:9.1f} {v2[-1] / 1e3:9.1f} {sum(v) / 1e6:9.1f}")
    # per stream: the longest gaps between consecutive kernels behind the start-up part (a coupling job that waits)
    t_lo2 = t0 + 0.4 * (ks[-1][1] - t0)
    bystream = collections.defaultdict(list)
    for s, e, k, st, wg, gr in ks:
        if s >= t_lo2:
            bystream[st].append((s, e, k))
    print("longest gaps between consecutive kernels of one stream (us, at ms, stream, before -> after):")
    gl = []
    for st, lst in bystream.items():
        lst.sort()
        for (s0, e0, k0), (s1, e1, k1) in zip(lst, lst[1:]):
            gl.append((s1 - e0, e0, st, k0, k1))
    for g, at, st, k0, k1 in sorted(gl, reverse=True)[:20]:
        print(f"  {g / 1e3:9.1f}  t={(at - t0) / 1e6:9.2f}  stream {st}  {k0} -> {k1}")
    print("longest kernel instances (start ms, dur us, name, stream, grid):")
    for s, e, k, st, wg, gr in sorted(ks, key=lambda r: -(r[1] - r[0]))[:25]:
        print(f"  t={(s - t0) / 1e6:9.2f}  {(e - s) / 1e3:9.1f}  {k:40s} stream {st} grid {gr}")
    # device idle gaps (no kernel resident on any stream)
    ev = []
    for s, e, k, st, wg, gr in ks:
        ev.append((s, 1, k)); ev.append((e, -1, k))
    ev.sort()
    depth = 0; last_end = None; last_k = None; gaps = []
    for t, dl, k in ev:
        if dl == 1:
            if depth == 0 and last_end is not None:
                gaps.append((t - last_end, last_end, last_k, k))
            depth += 1
        else:
            depth -= 1
            if depth == 0:
                last_end, last_k = t, k
    print("longest device-idle gaps (us, at ms, after kernel -> before kernel):")
    for g, at, a, b in sorted(gaps, reverse=True)[:15]:
        print(f"  {g / 1e3:9.1f}  t={(at - t0) / 1e6:9.2f}  {a} -> {b}")
    api = []
    for r in rows(d, "*hip_api_trace.csv"):
        api.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), int(r["Start_Timestamp"]), r["Function"], r.get("Thread_Id", "?")))
    if api:
        # (the first 40 % of the trace is start-up: module loads, first launches, workspace allocation)
        t_lo = t0 + 0.4 * (ks[-1][1] - t0)
        api = [a for a in api if a[1] >= t_lo]
        agg2 = collections.defaultdict(list)
        for du, s, f, th in api:
            agg2[f].append(du)
        print(f"{len(api)} HIP API calls; per function: calls avg_us max_us total_ms")
        for f, v in sorted(agg2.items(), key=lambda kv: -sum(kv[1]))[:20]:
            print(f"  {f:40s} {len(v):8d} {sum(v) / len(v) / 1e3:9.1f} {max(v) / 1e3:10.1f} {sum(v) / 1e6:9.1f}")
        print("longest HIP API calls (dur us, at ms, function, thread):")
        for du, s, f, th in sorted(api, reverse=True)[:25]:
            print(f"  {du / 1e3:10.1f}  t={(s - t0) / 1e6:9.2f}  {f}  thread {th}")


if __name__ == "__main__":
    main(sys.argv[1])
