#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_asg" -o t -- python "$R/scratch/asg_one.py" C3 3 > "$R/gpurun_out/asg_one.log" 2>&1
tail -4 "$R/gpurun_out/asg_one.log"
cd "$R"; f=$(find gpurun_out/trace_asg -name "*kernel_trace.csv" | head -1)
python scratch/trace_summary.py "$f" > gpurun_out/trace_summary.txt 2>&1; cat gpurun_out/trace_summary.txt | cut -c1-1500
rm -rf gpurun_out/trace_asg
