#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_a" -o t -- python "$R/scratch/asg_pool.py" 2 > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/trace_a -name "*kernel_trace.csv" | head -1); python scratch/trace_summary.py "$f" | tail -8 | cut -c1-1700; rm -f "$f"
