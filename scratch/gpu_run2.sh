#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name"; timeout "$TMO" "$@" > "$R/gpurun_out/$name.log" 2>&1; echo "rc=$? ($name)"; tail -n "${TAILN:-6}" "$R/gpurun_out/$name.log"; }
TMO=900 TAILN=15 run gpu_all python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider
cd /tmp
TMO=600 TAILN=3 run rocprof rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_r1" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline
cd "$R"; f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; [ -n "$f" ] && head -30 "$f"
