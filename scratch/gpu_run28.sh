#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_reference_suite.py -q --tb=short -x -k "ode or dopri or euler or mlp" -p no:cacheprovider 2>&1 | tail -3
echo "--- fused"; CFM_ODE_FUSED=1 timeout 200 python scratch/ode_check.py 2>&1 | grep -v amdgpu
echo "--- layer-per-kernel"; CFM_ODE_FUSED=0 timeout 200 python scratch/ode_check.py 2>&1 | grep -v amdgpu
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob('/tmp/traj*_1_*.npy')):
    a=np.load(f); b=np.load(f.replace('_1_','_0_',1))
    print(f.split('/')[-1], "max abs diff fused vs unfused:", float(np.abs(a-b).max()), "rel", float(np.abs(a-b).max()/np.abs(b).max()))
PY
