#!/bin/bash
# full check of HEAD: gpu tests, bench line, rocprof kernel stats + PMC passes of the same bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider ) 2>&1 | tail -5
( time timeout 400 python bench.py ) > gpurun_out/bench_v5.log 2>&1; grep real gpurun_out/bench_v5.log
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_v5" -o b -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_v5.log" 2>&1
cd "$R"; find gpurun_out/prof_v5 -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  cd /tmp; timeout 600 rocprofv3 --pmc $c --output-format csv -d "$R/gpurun_out/pmc_v5_$c" -o p -- python "$R/bench.py" --no-cpu-baseline --steps 6 --warmup 2 > "$R/gpurun_out/pmc_v5_$c.log" 2>&1
  cd "$R"; f=$(find gpurun_out/pmc_v5_$c -name "*counter_collection.csv" | head -1); python tools/pmc_summary.py "$f" 8 > gpurun_out/pmc_v5_$c.summary.csv; rm -f "$f"; head -6 gpurun_out/pmc_v5_$c.summary.csv
done
