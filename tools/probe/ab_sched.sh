#!/bin/bash
# Round-5 A/B of bench.py settings (run through gpurun): the epsilon reduction factor of the asynchronous auction in the pipelined
# loop, three interleaved passes.  Output kept in profiles/r5_async_sweep.txt.
cd /root/repo
rm -f gpurun_out/r5_theta.txt
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --repeats 9 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$*', 'ms_per_step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'seq', round(d['ms_per_step_sequential'],3))" >> gpurun_out/r5_theta.txt; }
for pass in 1 2 3; do
run
run --solver-sched 3,0,0,-1
run --solver-sched 2.5,0,0,-1
run --solver-sched 2,0,0,-1
done
sort -s -k1,2 gpurun_out/r5_theta.txt
