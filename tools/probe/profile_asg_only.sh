#!/bin/bash
# The solver part of tools/profile_round.sh alone (kernel trace + FETCH / WRITE passes of lone solves and of the batch form,
# kernel statistics of the default bench command): re-run after a change that touches only the exact solver.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O/raw
rocprofv3 --kernel-trace --output-format csv -d $O/raw/bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json.log 2>&1
python tools/prof_summary.py stats $O/raw/bench $O/bench_kernel_stats.csv
python tools/overlap_report.py $O/raw/bench > $O/bench_overlap.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/raw/asg_$C -- python tools/asg_trace.py run > $O/asg_$C.log 2>&1
  python tools/prof_summary.py pmc $O/raw/asg_$C $O/asg_pmc_$C.csv
done
python tools/prof_summary.py asgjson $O/asg_pmc_FETCH_SIZE.csv $O/asg_pmc_WRITE_SIZE.csv $O/asg_pmc_summary.json
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_trace -- python tools/asg_trace.py run > /dev/null 2>&1
python tools/asg_trace.py summary $O/raw/asg_trace > $O/asg_trace_summary.txt 2>&1
python tools/prof_summary.py stats $O/raw/asg_trace $O/asg_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_batch -- python tools/asg_batch_bench.py 4 > $O/asg_batch_bench.txt 2>&1
python tools/prof_summary.py stats $O/raw/asg_batch $O/asg_batch_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/raw/asg_batch_$C -- python tools/asg_batch_bench.py 4 > /dev/null 2>&1
  python tools/prof_summary.py pmc $O/raw/asg_batch_$C $O/asg_batch_pmc_$C.csv
done
rm -rf $O/raw
cat $O/asg_pmc_summary.json
