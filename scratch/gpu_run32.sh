#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 60 ./scratch/probe/bw_probe
echo "== default lib"; timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu
for v in b c d g h; do echo "== variant $v"; CFM_LIB_OVERRIDE=scratch/variants/sk_$v.so timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu; done
