#!/bin/bash
# full check of HEAD: gpu tests, bench line, rocprof kernel stats of the same bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider ) 2>&1 | tail -6
( time timeout 400 python bench.py ) > gpurun_out/bench_v4.log 2>&1; tail -2 gpurun_out/bench_v4.log | cut -c1-3000
cd /tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_v4" -o b -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_v4.log" 2>&1
cd "$R"; f=$(find gpurun_out/prof_v4 -name "*kernel_stats.csv" | head -1); cut -c1-160 "$f" | head -16; find gpurun_out/prof_v4 -name "*kernel_trace.csv" -delete
