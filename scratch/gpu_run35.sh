#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "== $1 CFM_SK_STREAM=$2"; CFM_SK_STREAM=$2 CFM_LIB_OVERRIDE=scratch/variants/sk_$1.so timeout 100 python scratch/sk_bench.py 2>&1 | grep -v amdgpu; }
run u4w8o2 1; run u4w8o2 2; run u4w4o3 3; run u4w4o3 2; run u8w4o3 3; run u8w4o3 2; run u4w8o4 2
