#!/bin/bash
# final check of HEAD: gpu tests, smoke, bench line, rocprof kernel stats of the same bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider ) 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py > gpurun_out/bench_v10.log 2>&1; grep "^{" gpurun_out/bench_v10.log | cut -c1-400
cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_v10" -o b -- python "$R/bench.py" --no-cpu-baseline > "$R/gpurun_out/prof_v10.log" 2>&1
cd "$R"; find gpurun_out/prof_v10 -name "*kernel_trace.csv" -delete; ls gpurun_out/prof_v10/*/ | head
