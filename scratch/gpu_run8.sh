#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_gpu_kernels.py -q --tb=short -x -k "assign" -p no:cacheprovider 2>&1 | tail -3
timeout 300 python scratch/asg_one.py C3 2 2>&1 | tail -5
