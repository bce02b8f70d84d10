import collections, csv, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    if not k.lstrip("void ").startswith("sk_"): continue
    n = max(calls[(k, c)] for c in d)
    print(k, "calls", n)
    for c, v in sorted(d.items()): print(f"   {c:28s} {v / n:14.1f} per call")
