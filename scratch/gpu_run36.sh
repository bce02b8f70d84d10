#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFM_SK_FUSED=S1 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu
CFM_LIB_OVERRIDE=scratch/variants/sk_head.so CFM_SK_FUSED=S0 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu | grep -v "^\[\|vs oracle"
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob('/tmp/sk_S1_*.npy')):
    a=np.load(f); b=np.load(f.replace('sk_S1_','sk_S0_'))
    print(f.split('/')[-1], "new vs HEAD: equal bits", bool(np.array_equal(a,b)), float(np.abs(a-b).max()))
PY
echo "== new default"; timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
echo "== HEAD"; CFM_LIB_OVERRIDE=scratch/variants/sk_head.so timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "sinkhorn or golden or unbalanced or wasserstein or sb_cfm" 2>&1 | tail -3
