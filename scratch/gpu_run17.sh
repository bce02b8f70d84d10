#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy" -p no:cacheprovider 2>&1 | tail -3
echo "--- default"; CHECK=1 timeout 200 python scratch/asg_pool.py 8 2>&1 | grep -v amdgpu.ids
timeout 100 python scratch/bid_prof.py 2>&1 | grep -v amdgpu | tail -1
cd /tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_a" -o t -- python "$R/scratch/asg_pool.py" 1 > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/trace_a -name "*kernel_trace.csv" | head -1); python scratch/trace_summary.py "$f" | tail -8 | cut -c1-900; rm -f "$f"
