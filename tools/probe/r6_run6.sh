#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1500 python -m pytest tests/test_gpu_transport.py -x -q -m gpu -s > gpurun_out/r6_t6.log 2>&1
echo "transport tests rc=$?" >> gpurun_out/r6_t6.log
timeout 900 python tools/transport_bench.py > gpurun_out/r6_transport_bench.txt 2>&1
BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_sweep6.txt
CFM_LIB_PATH=tools/probe/libcfm_prefetch.so BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v amdgpu.ids | sed 's/^/PREFETCH /' >> gpurun_out/r6_sweep6.txt
BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6_sweep6.txt
CFM_LIB_PATH=tools/probe/libcfm_prefetch.so BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -v amdgpu.ids | sed 's/^/PREFETCH /' >> gpurun_out/r6_sweep6.txt
tail -30 gpurun_out/r6_t6.log; cat gpurun_out/r6_transport_bench.txt gpurun_out/r6_sweep6.txt
