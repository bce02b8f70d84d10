#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; shift; echo "=== $name"; timeout "$TMO" "$@" > "$R/gpurun_out/$name.log" 2>&1; echo "rc=$? ($name)"; tail -n "${TAILN:-6}" "$R/gpurun_out/$name.log"; }
TMO=300 TAILN=8 run k_assign python -m pytest tests/test_gpu_kernels.py -q --tb=short -x -k "assign" -p no:cacheprovider
TMO=300 TAILN=12 run fullsize python -m pytest tests/test_gpu_fullsize.py -q --tb=short -s -k "c3_exact or c2_dense" -p no:cacheprovider
TMO=300 TAILN=40 run sweep python scratch/asg_sweep.py ${SWEEP:-C3,C2}
