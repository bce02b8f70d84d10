#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py -q --tb=short -x -k "assign or exact or scipy or rectangular" -p no:cacheprovider 2>&1 | tail -2
for e in 0.02 0.05 0.08; do echo "--- stop_early=$e"; STOPE=$e timeout 100 python scratch/asg_pool.py 8 2>&1 | grep -v amdgpu | awk '{s+=$3; n++; printf "%s ", $3} END {printf " | mean %.2f ms\n", s/n}'; done
for e in 0.02 0.05; do CFM_ASG_STOPE=$e timeout 200 python bench.py --steps 40 --no-cpu-baseline --no-sinkhorn 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop_early $e', round(d[\"value\"]), round(d[\"ms_per_step\"],3), round(d[\"assign_ms_per_step\"],3), d[\"ms_per_step_sequential\"])"; done
timeout 200 python scratch/asg_zoo.py 2>&1 | grep -v amdgpu | cut -c1-140
