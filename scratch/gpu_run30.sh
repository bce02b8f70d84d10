#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
CFM_SK_FUSED=1 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu
CFM_SK_FUSED=0 timeout 200 python scratch/sk_check.py 2>&1 | grep -v amdgpu
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob('/tmp/sk_1_*.npy')):
    a=np.load(f); b=np.load(f.replace('sk_1_','sk_0_'))
    print(f.split('/')[-1], "fused vs plain max rel diff:", float(np.abs(a-b).max()/np.abs(b).max()))
PY
timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
