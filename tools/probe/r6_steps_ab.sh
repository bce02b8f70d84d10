#!/bin/bash
# per-launch durations of the chip-wide steps of a lone C3 solve + the batch form's kernel statistics (round 6 A/B of single steps)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/sprof; mkdir -p $O/raw
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_trace -- python tools/asg_trace.py run > /dev/null 2>&1
python tools/asg_trace.py summary $O/raw/asg_trace > $O/asg_trace_summary.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_batch -- python tools/asg_batch_bench.py 4 > $O/asg_batch_bench.txt 2>&1
python tools/prof_summary.py stats $O/raw/asg_batch $O/asg_batch_kernel_stats.csv
rm -rf $O/raw
grep -A2 "per-launch" $O/asg_trace_summary.txt | tail -3; head -6 $O/asg_batch_kernel_stats.csv; grep median $O/asg_batch_bench.txt
