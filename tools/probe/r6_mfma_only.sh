#!/bin/bash
# the MFMA-busy part of tools/profile_round.sh alone
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof; mkdir -p $O/raw
rocprofv3 --kernel-trace --output-format csv -d $O/raw/mfma_trace -- python tools/mfma_probe.py > /dev/null 2>&1
python tools/prof_summary.py stats $O/raw/mfma_trace $O/mfma_kernel_stats.csv
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/raw/mfma_pmc -- python tools/mfma_probe.py > $O/mfma_pmc.log 2>&1
python tools/prof_summary.py pmc $O/raw/mfma_pmc $O/mfma_pmc.csv
python tools/prof_summary.py util $O/mfma_pmc.csv $O/mfma_util.csv
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/raw/mfma_pmc2 -- python tools/mfma_probe.py > $O/mfma_pmc2.log 2>&1
python tools/prof_summary.py pmc $O/raw/mfma_pmc2 $O/mfma_pmc2.csv
rm -rf $O/raw
cat $O/mfma_util.csv; cat $O/mfma_kernel_stats.csv | cut -c1-120
