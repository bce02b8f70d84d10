#!/bin/bash
# Quick look (run through gpurun): rocprofv3 kernel trace of the pipelined bench loop -> per-kernel statistics + overlap report.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_quick; rm -rf $O; mkdir -p $O/raw
rocprofv3 --kernel-trace --output-format csv -d $O/raw/bench -- python bench.py --steps 40 --warmup 5 --no-legs --no-cpu-baseline --repeats 3 > $O/bench_under_trace.json.log 2>&1
python tools/prof_summary.py stats $O/raw/bench $O/bench_kernel_stats.csv
python tools/overlap_report.py $O/raw/bench > $O/bench_overlap.txt 2>&1
rm -rf $O/raw
head -40 $O/bench_kernel_stats.csv; cat $O/bench_overlap.txt | head -40
