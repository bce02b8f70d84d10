#!/bin/bash
# Round-5 A/B of the exact solver (run through gpurun): tools/asg_async_ab.py over 40 instances, the solver tests, then bench.py
# regions for two settings of cfm_assign_set_async, interleaved.  Output: gpurun_out/r5_async_ab*.txt (kept in profiles/r5_async_sweep.txt).
cd /root/repo
NINST=16 BENCH_POOL=1 ASYNC_LIST="1,16,4;2,16,4;0,16,4;2,16,4" timeout 400 python tools/asg_async_ab.py 1 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r5_async_ab8.txt
timeout 900 python -m pytest tests/test_gpu_assign_batch.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_prefetch.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3 >> gpurun_out/r5_async_ab8.txt
for pass in 1 2; do
for cfg in "1,16,4" "2,16,4"; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --repeats 9 --solver-async $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('pass $pass async=$cfg ms_per_step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'seq', round(d['ms_per_step_sequential'],3))" >> gpurun_out/r5_async_ab8.txt
done
done
cat gpurun_out/r5_async_ab8.txt
