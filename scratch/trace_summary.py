import csv, sys, collections
rows=[]
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0]))
rows.sort()
# split into assign calls at asg_minmax
calls=[]; cur=None
for s,e,k in rows:
    if k.startswith("asg_minmax"): cur=[]; calls.append(cur)
    if cur is not None and k.startswith("asg_"): cur.append((s,e,k))
for ci,c in enumerate(calls):
    span=(c[-1][1]-c[0][0])/1e3
    busy=sum(e-s for s,e,k in c)/1e3
    by=collections.defaultdict(lambda:[0,0.0])
    for s,e,k in c: by[k][0]+=1; by[k][1]+=(e-s)/1e3
    big=sorted(((e-s)/1e3,k,idx) for idx,(s,e,k) in enumerate(c))[-6:]
    gaps=[(c[i+1][0]-c[i][1])/1e3 for i in range(len(c)-1)]
    print(f"call {ci}: kernels={len(c)} span={span:.1f}us busy={busy:.1f}us gaps_total={sum(gaps):.1f}us max_gap={max(gaps):.1f}us")
    for k,(n,t) in by.items(): print(f"    {k:12s} n={n:5d} total={t:9.1f}us avg={t/n:7.2f}us")
    print("    biggest:", [(round(d,1),k,idx) for d,k,idx in big])
    # duration profile of the first 160 wide kernels
    w=[round((e-s)/1e3,1) for s,e,k in c if k.startswith("asg_wide")][:150]
    print("    wide durations:", w)
    ct=[round((e-s)/1e3,1) for s,e,k in c if k.startswith("asg_ctrl")][:150]
    print("    ctrl durations:", ct)
    biggaps=sorted(((g,i) for i,g in enumerate(gaps)),reverse=True)[:5]
    print("    biggest gaps (us, after kernel idx):", [(round(g,1),i,c[i][2]) for g,i in biggaps])
