#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy or rectangular" -p no:cacheprovider 2>&1 | tail -3
echo "--- default stream (no graph)"; CHECK=1 timeout 100 python scratch/asg_pool.py 8 2>&1 | grep -v amdgpu | cut -c1-60
echo "--- side stream (graph)"; SIDE=1 CHECK=1 timeout 100 python scratch/asg_pool.py 8 2>&1 | grep -v amdgpu | cut -c1-60
for p in 0 1 2 3 4 6; do timeout 200 python bench.py --pipeline $p --steps 40 --no-cpu-baseline --no-sinkhorn 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d[\"config\"][\"schedule\"][:34], round(d[\"value\"]), round(d[\"ms_per_step\"],3), round(d[\"assign_ms_per_step\"],3), d[\"ms_per_step_sequential\"])"; done
