#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python -m pytest tests/test_gpu_unbalanced.py -q --tb=short -p no:cacheprovider 2>&1 | tail -3
echo "--- base (rpw4 occ3)"; timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
for v in rpw4o4 rpw8 rpw8o3 rpw16; do echo "--- $v"; CFM_LIB_OVERRIDE=$PWD/scratch/variants/$v.so timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu; done
