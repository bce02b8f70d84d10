#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy" -p no:cacheprovider 2>&1 | tail -8
echo "--- handoff 6, arr 30"; CHECK=1 timeout 200 python scratch/asg_pool.py 8 6 30 2>&1 | grep -v amdgpu.ids
echo "--- handoff 6, arr 15"; timeout 200 python scratch/asg_pool.py 8 6 15 2>&1 | grep -v amdgpu.ids
echo "--- handoff 0 (dense multi-source only), arr 15"; timeout 200 python scratch/asg_pool.py 4 0 15 2>&1 | grep -v amdgpu.ids
echo "--- handoff 100 (sparse only), arr 30"; timeout 200 python scratch/asg_pool.py 8 100 30 2>&1 | grep -v amdgpu.ids
