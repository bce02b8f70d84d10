#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/r6_pass2.txt
for rep in 1 2; do
BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v "amdgpu.ids\|identical" | sed 's/^/MAIN  /' >> gpurun_out/r6_pass2.txt
CFM_LIB_PATH=tools/probe/libcfm_pass2.so BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v "amdgpu.ids\|identical" | sed 's/^/PASS2 /' >> gpurun_out/r6_pass2.txt
done
CFM_LIB_PATH=tools/probe/libcfm_pass2.so timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_assign_batch.py -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r6_pass2.txt
cat gpurun_out/r6_pass2.txt
