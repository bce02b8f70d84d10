cd /root/repo
rm -f gpurun_out/r5_sched.txt
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --repeats 9 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$*', 'ms_per_step', round(d['ms_per_step'],4), 'min', round(d['ms_per_step_min'],4), 'steady', 'n/a' if not d.get('steady_state') else round(d['steady_state']['ms_per_step'],4))" >> gpurun_out/r5_sched.txt; }
for pass in 1 2; do
run
run --solver-async 2,16,4,32
run --solver-async 2,16,4,128
run --group 2 --tail 1
run --group 3 --tail 2,1
run --group 6 --tail 3,2,1
run --pipeline 2
run --pipeline 4
run --solver-async 2,32,4
done
sort -s -k1,6 gpurun_out/r5_sched.txt
