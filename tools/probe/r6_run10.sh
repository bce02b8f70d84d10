#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/probe/blocking_sync_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r6_blocking_probe.txt
for B in 1 0 1 0; do python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --blocking-sync $B 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('blocking_sync=$B ms/step', round(d['ms_per_step'],4), 'p95', round(d['ms_per_step_p95'],3), 'max', round(d['ms_per_step_max'],3), 'host cpu ms/step', round(d['host_cpu_ms_per_step'],3), 'threads', d['host_threads_busy_fraction'][:5])" >> gpurun_out/r6_blocking_probe.txt; done
cat gpurun_out/r6_blocking_probe.txt
