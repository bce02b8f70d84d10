#!/bin/bash
# Round 6: the FIRST jobs of the prefetch schedule (--ramp), regions of 20 steps (the driver's arguments), 21 regions each, one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/r6_ramp_sweep.txt
for R in "1" "1,2,3" "1,2" "1,1,2" "2" "1" "1,1" "2,3" "1,2,4" "1,3" "1,2,3"; do
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --repeats 21 --ramp "$R" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('ramp [$R]: ms/step', round(d['ms_per_step'],4), 'p95', round(d['ms_per_step_p95'],3), 'max', round(d['ms_per_step_max'],3), 'jobs', d['config']['prefetch_job_sizes'])" >> gpurun_out/r6_ramp_sweep.txt
done
cat gpurun_out/r6_ramp_sweep.txt
