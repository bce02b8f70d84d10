#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name"; timeout "$TMO" "$@" > "$R/gpurun_out/$name.log" 2>&1; echo "rc=$? ($name)"; tail -n "${TAILN:-6}" "$R/gpurun_out/$name.log"; }
TMO=900 TAILN=4 run gpu_all python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider
TMO=300 TAILN=2 run bench python bench.py
cd /tmp
TMO=300 TAILN=2 run rocprof rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_r1b" -o bench -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline
TMO=300 TAILN=2 run pmc_fetch rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_fetch" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline
TMO=300 TAILN=2 run pmc_write rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_write" -o bench -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline
cd "$R"; find gpurun_out/prof_r1b gpurun_out/pmc_fetch gpurun_out/pmc_write -type f | head -20
f=$(find gpurun_out/prof_r1b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 "$f" | head -14
# keep the pulled payload small: counter CSVs can be large
for d in gpurun_out/pmc_fetch gpurun_out/pmc_write; do find $d -name "*counter_collection.csv" -size +20M -exec sh -c 'head -c 20000000 "$1" > "$1.trunc" && rm "$1"' _ {} \; ; done
find gpurun_out/prof_r1b -name "*kernel_trace.csv" -size +20M -delete
du -sh gpurun_out
