#!/bin/bash
# full GPU suite + smoke + default bench (what the driver runs at round end)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/r6_gputest_full.log 2>&1
echo "rc=$?" >> gpurun_out/r6_gputest_full.log
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r6_gputest_full.log 2>&1
python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err
tail -5 gpurun_out/r6_gputest_full.log
