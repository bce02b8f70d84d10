#!/bin/bash
# Round 6, GPU run 4: the new transportation solver (tests + timings), the auction after the bid prefetch fix
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_gpu_transport.py -x -q -m gpu -s > gpurun_out/r6_t4.log 2>&1
echo "transport tests rc=$?" >> gpurun_out/r6_t4.log
timeout 600 python -m pytest tests/test_gpu_assign_batch.py tests/test_gpu_kernels.py -x -q -m gpu >> gpurun_out/r6_t4.log 2>&1
echo "assign tests rc=$?" >> gpurun_out/r6_t4.log
timeout 300 python tools/transport_bench.py > gpurun_out/r6_transport_bench.txt 2>&1
BENCH_POOL=1 NINST=16 timeout 300 python tools/asg_sched_sweep.py "theta=2.5" "theta=2.5" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_sweep4.txt
tail -25 gpurun_out/r6_t4.log; cat gpurun_out/r6_transport_bench.txt gpurun_out/r6_sweep4.txt
