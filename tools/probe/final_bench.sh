cd /root/repo
timeout 300 python -m pytest tests/test_gpu_assign_batch.py tests/test_gpu_prefetch.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
python -c "
import json; d=json.load(open('gpurun_out/r5_bench_default.json')); r=d['roofline']; print(d['ms_per_step'], d['ms_per_step_all'], d['ms_per_step_sequential'], d['steady_state']['ms_per_step'], r['solve_ms'], r['frac'], r['batch']['frac'], d['value_public_api']['ms_per_step_pipelined'], d['value_public_api']['ms_per_step_pipelined_all'])"
