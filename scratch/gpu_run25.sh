#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_p" -o t -- python "$R/bench.py" --pipeline 3 --steps 30 --no-cpu-baseline --no-sinkhorn > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/trace_p -name "*kernel_trace.csv" | head -1); python scratch/trace_overlap.py "$f"; rm -f "$f"
