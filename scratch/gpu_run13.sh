#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
echo "--- stop 0.05 arr 15"; timeout 200 python scratch/asg_pool.py 8 6 15 0.05 2>&1 | grep -v amdgpu.ids
echo "--- stop 0.05 arr 15 elast 1e-4"; timeout 200 python scratch/asg_pool.py 8 6 15 0.05 5 1e-4 2>&1 | grep -v amdgpu.ids
echo "--- stop 0.1 arr 15"; timeout 200 python scratch/asg_pool.py 8 6 15 0.1 2>&1 | grep -v amdgpu.ids
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_a" -o t -- python "$R/scratch/asg_pool.py" 1 6 15 > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/trace_a -name "*kernel_trace.csv" | head -1); python scratch/trace_summary.py "$f" | tail -12 | cut -c1-2500; rm -f "$f"
