#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy" -p no:cacheprovider 2>&1 | tail -3
echo "--- default"; CHECK=1 timeout 200 python scratch/asg_pool.py 8 2>&1 | grep -v amdgpu.ids
timeout 300 python scratch/asg_zoo.py 2>&1 | grep -v amdgpu
