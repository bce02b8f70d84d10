#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy" -p no:cacheprovider 2>&1 | tail -3
echo "--- handoff 6"; CHECK=1 timeout 200 python scratch/asg_pool.py 8 6 2>&1 | grep -v amdgpu.ids
echo "--- handoff 3"; timeout 200 python scratch/asg_pool.py 8 3 2>&1 | grep -v amdgpu.ids
echo "--- handoff 10"; timeout 200 python scratch/asg_pool.py 8 10 2>&1 | grep -v amdgpu.ids
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_a" -o t -- python "$R/scratch/asg_pool.py" 1 6 > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/trace_a -name "*kernel_trace.csv" | head -1); python scratch/trace_summary.py "$f" | tail -8 | cut -c1-1500; rm -f "$f"
