#!/bin/bash
# asg_build A/B (round 6: the row strip of the list build in registers): kernel statistics of lone solves and of the batch form
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/bprof; mkdir -p $O/raw
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_trace -- python tools/asg_trace.py run > /dev/null 2>&1
python tools/prof_summary.py stats $O/raw/asg_trace $O/asg_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_batch -- python tools/asg_batch_bench.py 4 > $O/asg_batch_bench.txt 2>&1
python tools/prof_summary.py stats $O/raw/asg_batch $O/asg_batch_kernel_stats.csv
rm -rf $O/raw
head -6 $O/asg_kernel_stats.csv; head -6 $O/asg_batch_kernel_stats.csv; grep median $O/asg_batch_bench.txt
