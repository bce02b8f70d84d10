#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_fullsize.py -q --tb=short -x -k "sinkhorn or sb or dense or c2 or c5" -p no:cacheprovider 2>&1 | tail -4
timeout 300 python scratch/sk_bench.py 2>&1 | tail -3
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_sk" -o sk -- python "$R/scratch/sk_bench.py" > /dev/null 2>&1
cd "$R"; f=$(find gpurun_out/prof_sk -name "*kernel_stats.csv" | head -1); cut -c1-140 "$f" | head -8; find gpurun_out/prof_sk -name "*kernel_trace.csv" -delete
