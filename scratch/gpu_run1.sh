#!/bin/bash
# First GPU pass: smoke, per-kernel parity groups (separate processes so a fault in one group
# does not hide the others), reference suite, goldens, full-size, bench, rocprof.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showproductname 2>/dev/null | head -8 > gpurun_out/gpu.txt
run() { name=$1; shift; echo "=== $name"; timeout "$TMO" "$@" > gpurun_out/$name.log 2>&1; echo "rc=$? ($name)"; tail -n "${TAILN:-6}" gpurun_out/$name.log; }
TMO=300 run smoke python -c "import __graft_entry__ as g; g.smoke()"
for grp in cost sinkhorn assign sample xt_ut gather mlp ode; do
  TMO=600 TAILN=12 run k_$grp python -m pytest tests/test_gpu_kernels.py -q --tb=short -k "$grp" -p no:cacheprovider
done
TMO=600 TAILN=12 run refsuite python -m pytest tests/test_gpu_reference_suite.py -q --tb=short -p no:cacheprovider
TMO=600 TAILN=12 run golden python -m pytest tests/test_gpu_golden.py -q --tb=short -p no:cacheprovider
TMO=900 TAILN=25 run fullsize python -m pytest tests/test_gpu_fullsize.py -q --tb=short -s -p no:cacheprovider
TMO=600 TAILN=5 run bench python bench.py --steps 5 --warmup 2
TMO=300 TAILN=5 run bench_coupling python bench.py --steps 5 --warmup 2 --mode coupling --no-cpu-baseline --no-sinkhorn
cd /tmp && TMO=600 TAILN=3 run rocprof rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline
cd "$GRAFT_REPO_ROOT"; find gpurun_out/prof_r1 -name "*kernel_stats*" | head; f=$(find gpurun_out/prof_r1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
