#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel (calls, total, per-call).

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide
coalesced read (MI355X_MICROARCH.md, HBM section; calibrated here on sk_col_pass, which reads the
64 MiB cost matrix once and reports 32 940 KiB) -> the `x2` column is the corrected figure."""
import collections
import csv
import sys


def main(path, top=14):
    agg = collections.defaultdict(lambda: [0, 0.0])
    name = "?"
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][:48]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
            name = r["Counter_Name"]
    print(f"counter,{name},unit,KiB")
    print("kernel,calls,total_KiB,per_call_KiB,per_call_KiB_x2")
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"\"{k}\",{n},{v:.1f},{v / n:.2f},{2 * v / n:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 14)
