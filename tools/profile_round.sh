#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel stats of the default bench command, HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the
#   exact-assignment solves, MFMA-busy counters of the MFMA-bearing kernels.  Summaries land in gpurun_out/prof/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; rm -rf $O; mkdir -p $O/raw
rocprofv3 --kernel-trace --output-format csv -d $O/raw/bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_trace.json.log 2>&1
python tools/prof_summary.py stats $O/raw/bench $O/bench_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/raw/asg_$C -- python tools/asg_trace.py run > $O/asg_$C.log 2>&1
  python tools/prof_summary.py pmc $O/raw/asg_$C $O/asg_pmc_$C.csv
done
python tools/prof_summary.py asgjson $O/asg_pmc_FETCH_SIZE.csv $O/asg_pmc_WRITE_SIZE.csv $O/asg_pmc_summary.json
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_trace -- python tools/asg_trace.py run > /dev/null 2>&1
python tools/asg_trace.py summary $O/raw/asg_trace > $O/asg_trace_summary.txt 2>&1
python tools/prof_summary.py stats $O/raw/asg_trace $O/asg_kernel_stats.csv
# the batch form of the solver (what the pipelined schedule runs): cfm_assign_exact_batch_f32 with 4 problems
rocprofv3 --kernel-trace --output-format csv -d $O/raw/asg_batch -- python tools/asg_batch_bench.py 4 > $O/asg_batch_bench.txt 2>&1
python tools/prof_summary.py stats $O/raw/asg_batch $O/asg_batch_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/raw/asg_batch_$C -- python tools/asg_batch_bench.py 4 > /dev/null 2>&1
  python tools/prof_summary.py pmc $O/raw/asg_batch_$C $O/asg_batch_pmc_$C.csv
done
rocprofv3 --kernel-trace --output-format csv -d $O/raw/mfma_trace -- python tools/mfma_probe.py > /dev/null 2>&1
python tools/prof_summary.py stats $O/raw/mfma_trace $O/mfma_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/raw/mfma_pmc -- python tools/mfma_probe.py > $O/mfma_pmc.log 2>&1
python tools/prof_summary.py pmc $O/raw/mfma_pmc $O/mfma_pmc.csv
python tools/prof_summary.py util $O/mfma_pmc.csv $O/mfma_util.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/raw/mfma_pmc2 -- python tools/mfma_probe.py > $O/mfma_pmc2.log 2>&1
python tools/prof_summary.py pmc $O/raw/mfma_pmc2 $O/mfma_pmc2.csv
# kernel-space entropic solvers (unbalanced / partial): per-kernel statistics of the fused loops
rocprofv3 --kernel-trace --output-format csv -d $O/raw/ub -- python tools/ub_probe.py > /dev/null 2>&1
python tools/prof_summary.py stats $O/raw/ub $O/ub_kernel_stats.csv
# how the couplings in flight overlap with the model step in the pipelined loop
python tools/overlap_report.py $O/raw/bench > $O/bench_overlap.txt 2>&1
# Sinkhorn: per-config kernel statistics and HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), VALU-busy of variant B
for CFG in C5 C2; do
  rocprofv3 --kernel-trace --output-format csv -d $O/raw/sk_$CFG -- python tools/sk_probe.py $CFG > /dev/null 2>&1
  python tools/prof_summary.py stats $O/raw/sk_$CFG $O/sk_${CFG}_kernel_stats.csv
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/raw/sk_${CFG}_$C -- python tools/sk_probe.py $CFG > /dev/null 2>&1
    python tools/prof_summary.py pmc $O/raw/sk_${CFG}_$C $O/sk_${CFG}_pmc_$C.csv
  done
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/raw/sk_C2_valu -- python tools/sk_probe.py C2 > /dev/null 2>&1
python tools/prof_summary.py pmc $O/raw/sk_C2_valu $O/sk_C2_pmc_valu.csv
python tools/prof_summary.py skjson $O $O/sk_pmc_summary.json
# the C5 Sinkhorn leg window by window, with the solver's own state after each (iterations done, fp64-exp regime)
python tools/sk_windows.py C5 > $O/sk_C5_windows.txt 2>&1
python tools/sk_windows.py C2 >> $O/sk_C5_windows.txt 2>&1
rocprofv3 -L 2>/dev/null | grep -i -E "mfma|FETCH_SIZE|WRITE_SIZE|MfmaUtil" | head -40 > $O/counters_available.txt
rm -rf $O/raw
ls -la $O
