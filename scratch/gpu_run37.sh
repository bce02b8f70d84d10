#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
( time timeout 1200 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider ) 2>&1 | tail -6
timeout 400 python bench.py > gpurun_out/bench_v9.log 2>&1; grep "^{" gpurun_out/bench_v9.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['value','ms_per_step','assign_ms_per_step','ms_per_step_sequential','sinkhorn_iters_per_s']}); print(d['roofline']['frac'], d['roofline_sinkhorn']['frac'])"
