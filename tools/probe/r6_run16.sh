#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_train.py tests/test_gpu_rccl_world1.py tests/test_gpu_prefetch.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r6_fold.txt
for i in 1 2; do python tools/gemm_quick.py 2>&1 | grep -v amdgpu >> gpurun_out/r6_fold.txt; done
cat gpurun_out/r6_fold.txt
