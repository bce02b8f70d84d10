#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== default"; timeout 200 python scratch/asg_phases.py 2>&1 | grep -v amdgpu
for v in asg_nosplit asg_split1k asg_y2; do echo "== $v"; CFM_LIB_OVERRIDE=scratch/variants/$v.so timeout 200 python scratch/asg_phases.py 2>&1 | grep -v amdgpu | grep "mean solve\|SAP(ms"; done
