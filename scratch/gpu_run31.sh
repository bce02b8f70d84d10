#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFM_SK_FOLD=1 CFM_SK_FUSED=F1 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu
CFM_SK_FOLD=0 CFM_SK_FUSED=F0 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob('/tmp/sk_F1_*.npy')):
    a=np.load(f); b=np.load(f.replace('sk_F1_','sk_F0_'))
    print(f.split('/')[-1], "fold vs separate: equal bits", bool(np.array_equal(a,b)), float(np.abs(a-b).max()))
PY
echo "== default lib, fold=0"; CFM_SK_FOLD=0 timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
echo "== default lib, fold=1"; CFM_SK_FOLD=1 timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
for v in b c d; do echo "== variant $v fold=1"; CFM_LIB_OVERRIDE=scratch/variants/sk_$v.so timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu; done
