#!/bin/bash
# Round 6: the prefetch schedule on the new solver (workers x group), 21 regions of 20 steps each
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/r6_sched_sweep.txt
for P in 2 3 4; do for G in 2 4 6 8; do
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --repeats 21 --pipeline $P --group $G 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('pipeline $P group $G: ms/step', round(d['ms_per_step'],4), 'p95', round(d['ms_per_step_p95'],3), 'max', round(d['ms_per_step_max'],3), 'host cpu', round(d['host_cpu_ms_per_step'],2))" >> gpurun_out/r6_sched_sweep.txt
done; done
for T in "2,1" "1" "" "4,2,1"; do
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --repeats 21 --tail "$T" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('pipeline 3 group 4 tail [$T]: ms/step', round(d['ms_per_step'],4), 'p95', round(d['ms_per_step_p95'],3), 'max', round(d['ms_per_step_max'],3))" >> gpurun_out/r6_sched_sweep.txt
done
cat gpurun_out/r6_sched_sweep.txt
