import csv, sys, collections
rows=[]
with open(sys.argv[1]) as f:
    rd=csv.DictReader(f)
    cols=rd.fieldnames
    for r in rd:
        rows.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"].split("(")[0][:40],r.get("Queue_Id","?"),r.get("Stream_Id","?")))
print("columns:",cols)
rows.sort()
# restrict to the last 60% of the trace (steady state)
t0=rows[0][0]; t1=rows[-1][1]; lo=t0+0.5*(t1-t0); hi=t0+0.9*(t1-t0)
sel=[r for r in rows if r[0]>=lo and r[1]<=hi]
span=(hi-lo)/1e3
tot=sum(e-s for s,e,_,_,_ in sel)/1e3
# union
ev=[]
for s,e,_,_,_ in sel: ev.append((s,1)); ev.append((e,-1))
ev.sort(); cur=0; last=None; union=0; conc=collections.Counter()
for t,d in ev:
    if last is not None and cur>0: union+=t-last
    if last is not None: conc[cur]+=t-last
    cur+=d; last=t
print(f"window {span:.0f} us: sum of kernel durations {tot:.0f} us, union busy {union/1e3:.0f} us ({union/1e3/span*100:.1f}% of wall), avg concurrency when busy {tot/(union/1e3):.2f}")
print("time share by number of concurrent kernels:",{k:round(v/1e3/span*100,1) for k,v in sorted(conc.items())})
by=collections.defaultdict(lambda:[0,0.0])
for s,e,k,q,st in sel: by[k][0]+=1; by[k][1]+=(e-s)/1e3
for k,(n,t) in sorted(by.items(),key=lambda kv:-kv[1][1])[:8]: print(f"  {k:40s} n={n:6d} total={t:9.0f}us avg={t/n:7.2f}us")
qs=collections.Counter((q,st) for _,_,_,q,st in sel); print("queues/streams:",dict(qs))
