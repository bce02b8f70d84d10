"""Run a few exact-assignment solves (C3: B=4096, d=784) for a rocprofv3 --kernel-trace capture, or
summarise such a capture: per-launch kernel durations and the gaps between launches of one solve.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/asg_trace -- python tools/asg_trace.py run
    python tools/asg_trace.py summary gpurun_out/asg_trace

Test / measurement infrastructure; not part of the product path."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run():
    import torch
    import cfm_amd  # noqa: F401
    from cfm_amd import _lib
    import cfm_amd.optimal_transport as ot
    import bench
    _lib.load(); dev = _lib.require_gpu()
    with torch.cuda.stream(torch.cuda.Stream()):
        Ms = [ot.cost_matrix(x0, x1) for (x0, x1) in bench.synth_batches(4096, 784, 4, 1000, dev)]
        for rep in range(2):
            for M in Ms:
                ot.assign_exact(M)
        torch.cuda.synchronize()


def summary(d):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    asg = [(s, e, k) for (s, e, k) in rows if k.startswith("asg_") or "asg_" in k]
    # split into solves at asg_init
    solves, cur = [], []
    for s, e, k in asg:
        if "asg_init" in k:
            if cur:
                solves.append(cur)
            cur = []
        cur.append((s, e, k))
    if cur:
        solves.append(cur)
    print(f"{len(solves)} solves traced")
    for si, sv in enumerate(solves[-4:]):
        import collections
        dur = collections.defaultdict(list); gaps = []
        for q, (s, e, k) in enumerate(sv):
            dur[k.split("(")[0]].append((e - s) / 1e3)
            if q:
                gaps.append((s - sv[q - 1][1]) / 1e3)
        total = (sv[-1][1] - sv[0][0]) / 1e3
        print(f"solve {si}: {len(sv)} launches, first start -> last end {total:.0f} us, sum of gaps {sum(gaps):.0f} us "
              f"(median gap {sorted(gaps)[len(gaps)//2]:.2f} us)")
        for k, v in dur.items():
            v2 = sorted(v)
            print(f"   {k:12s} n={len(v):4d} sum {sum(v):8.1f} us  median {v2[len(v2)//2]:6.2f}  p10 {v2[len(v2)//10]:6.2f}  p90 {v2[(9*len(v2))//10]:6.2f}  max {v2[-1]:7.2f}")
        if si == len(solves[-4:]) - 1:
            print("   per-launch durations (us) of asg_step, in order:")
            print("   " + " ".join(f"{(e - s)/1e3:.1f}" for (s, e, k) in sv if "asg_step" in k))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summary(sys.argv[2])
