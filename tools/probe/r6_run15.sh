#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -f gpurun_out/r6_farmult.txt
for rep in 1 2; do
BENCH_POOL=1 NINST=16 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v "amdgpu.ids\|identical" | sed 's/^/FAR 3.0 /' >> gpurun_out/r6_farmult.txt
for v in 1.0 1.5 2.0 4.0; do CFM_LIB_PATH=tools/probe/libcfm_far$v.so BENCH_POOL=1 NINST=16 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v "amdgpu.ids\|identical" | sed "s/^/FAR $v /" >> gpurun_out/r6_farmult.txt; done
done
cat gpurun_out/r6_farmult.txt
