cd /root/repo
export TMPDIR=/tmp
python -m pytest tests/test_gpu_partition.py tests/test_gpu_assign_batch.py tests/test_gpu_rows_sampling.py tests/test_gpu_rccl_world1.py -x -q -m gpu > gpurun_out/r6_t1.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_t1.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench1.json 2> gpurun_out/r6_bench1.err
BENCH_POOL=1 NINST=16 python tools/asg_sched_sweep.py "theta=2.5" "arr=20" "arr=40" "last_div=8" "last_div=16" "last_div=64" "eps_last=1e-7" "eps_last=1e-8" "stop=0.01" "stop=0.005" "theta=2.0" "theta=1.7" "last_div=16,arr=20" "last_div=64,arr=40" "eps_last=1e-7,last_div=16,arr=20" "theta=2.0,last_div=16,arr=20" "eps0=4e-3" "eps0=2e-3,theta=2.0" "stop=0.01,last_div=8,arr=20" "theta=2.5" > gpurun_out/r6_sweep1.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --priority -1 > gpurun_out/r6_bench_prio.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --blocking-sync 0 > gpurun_out/r6_bench_spin.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline > gpurun_out/r6_bench_again.json 2>/dev/null
