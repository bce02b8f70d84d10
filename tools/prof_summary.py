"""Summarise rocprofv3 output directories into the small files kept under profiles/.
    python tools/prof_summary.py stats  <dir> <out.csv>        per-kernel calls / total / avg / min / max (kernel trace)
    python tools/prof_summary.py pmc    <dir> <out.csv>        per-kernel per-counter mean per call
Measurement infrastructure."""
import collections, csv, glob, os, sys


def _rows(d, pat):
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


def stats(d, out):
    agg = collections.defaultdict(list)
    for r in _rows(d, "*kernel_trace.csv"):
        agg[r["Kernel_Name"].split("(")[0][:80]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    with open(out, "w") as fh:
        fh.write("kernel,calls,total_us,avg_us,min_us,max_us,percent\n")
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f"\"{k}\",{len(v)},{sum(v)/1e3:.1f},{sum(v)/len(v)/1e3:.2f},{min(v)/1e3:.2f},{max(v)/1e3:.2f},{100*sum(v)/tot:.2f}\n")


def pmc(d, out):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in _rows(d, "*counter_collection.csv"):
        k = r["Kernel_Name"].split("(")[0][:80]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    with open(out, "w") as fh:
        fh.write("kernel,counter,calls,mean_per_call\n")
        for k, dct in sorted(agg.items()):
            for c, v in sorted(dct.items()):
                fh.write(f"\"{k}\",{c},{calls[(k, c)]},{v / calls[(k, c)]:.3f}\n")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
