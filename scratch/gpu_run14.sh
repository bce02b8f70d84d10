#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy" -p no:cacheprovider 2>&1 | tail -3
echo "--- q=0.75 stop 0.02 arr 15"; CHECK=1 timeout 200 python scratch/asg_pool.py 8 6 15 2>&1 | grep -v amdgpu.ids
echo "--- q=1.0"; MSQ=1.0 timeout 200 python scratch/asg_pool.py 8 6 15 2>&1 | grep -v amdgpu.ids
echo "--- q=0.5"; MSQ=0.5 timeout 200 python scratch/asg_pool.py 8 6 15 2>&1 | grep -v amdgpu.ids
echo "--- q=0.75 stop 0.05"; timeout 200 python scratch/asg_pool.py 8 6 15 0.05 2>&1 | grep -v amdgpu.ids
echo "--- q=0.75 stop 0.1"; timeout 200 python scratch/asg_pool.py 8 6 15 0.1 2>&1 | grep -v amdgpu.ids
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_a" -o t -- python "$R/scratch/asg_pool.py" 1 6 15 > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/trace_a -name "*kernel_trace.csv" | head -1); python scratch/trace_summary.py "$f" | tail -7 | cut -c1-1800; rm -f "$f"
