#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R="$PWD"; export TMPDIR=/tmp
for st in 0 1; do
cd /tmp; CFM_SK_STREAM=$st timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/gpurun_out/pmc_sk$st" -o p -- python "$R/scratch/sk_bench.py" > "$R/gpurun_out/pmc_sk$st.log" 2>&1
cd "$R"; f=$(find gpurun_out/pmc_sk$st -name "*counter_collection.csv" | head -1); echo "== CFM_SK_STREAM=$st"; python scratch/pmc_multi.py "$f"; rm -f "$f"
done
