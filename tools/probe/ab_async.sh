cd /root/repo
rm -f gpurun_out/r5_async_ab5.txt
for pass in 1 2 3; do
for cfg in "0,0,1" "1,16,1" "1,16,4" "1,24,4" "1,32,4" "1,16,8"; do
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --repeats 9 --solver-async $cfg 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('pass $pass async=$cfg ms_per_step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'seq', round(d['ms_per_step_sequential'],3))" >> gpurun_out/r5_async_ab5.txt
done
done
sort -k3,3 -s gpurun_out/r5_async_ab5.txt
