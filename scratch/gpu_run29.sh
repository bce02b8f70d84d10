#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_ode" -o b -- python "$R/scratch/ode_bench.py" > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/prof_ode -name "*kernel_stats.csv" | head -1); cut -c1-150 "$f" | head -8; find gpurun_out/prof_ode -name "*kernel_trace.csv" -delete
