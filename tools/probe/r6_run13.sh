#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" "theta=2.5" 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r6_sweep13.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_assign_batch.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/r6_sweep13.txt
python tools/asg_report.py 2>&1 | tail -15 >> gpurun_out/r6_sweep13.txt
cat gpurun_out/r6_sweep13.txt
