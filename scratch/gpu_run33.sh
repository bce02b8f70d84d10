#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFM_SK_STREAM=1 CFM_SK_FUSED=S1 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu
CFM_SK_STREAM=0 CFM_SK_FUSED=S0 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu | grep -v "^\[\|vs oracle"
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob('/tmp/sk_S1_*.npy')):
    a=np.load(f); b=np.load(f.replace('sk_S1_','sk_S0_'))
    print(f.split('/')[-1], "stream vs one-shot: equal bits", bool(np.array_equal(a,b)), float(np.abs(a-b).max()))
PY
for g in 0 1 2; do echo "== default lib (8 waves), CFM_SK_STREAM=$g"; CFM_SK_STREAM=$g timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu; done
for g in 2 4; do echo "== 4 waves, CFM_SK_STREAM=$g"; CFM_SK_STREAM=$g CFM_LIB_OVERRIDE=scratch/variants/sk_s4.so timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu; done
