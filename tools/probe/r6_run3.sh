#!/bin/bash
# Round 6, GPU run 3: the auction with claimed / adopted row groups — solver tests, lone-solve sweep line, full bench (51 regions)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_assign_batch.py tests/test_gpu_fullsize.py tests/test_gpu_partition.py tests/test_gpu_prefetch.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r6_t3.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_t3.log
BENCH_POOL=1 NINST=16 python tools/asg_sched_sweep.py "theta=2.5" "theta=2.5" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_sweep3.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench3.json 2> gpurun_out/r6_bench3.err
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r6_bench3b.json 2>/dev/null
tail -3 gpurun_out/r6_t3.log; cat gpurun_out/r6_sweep3.txt
