#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "== default (tail poll, build vec, relax prefetch)"; timeout 200 python scratch/asg_phases.py 2>&1 | grep -v amdgpu | tail -16
[ "${PIPESTATUS[0]}" = "0" ] || { echo FAILED; exit 1; }
echo "== CFM_ASG_TAILPOLL=0"; CFM_ASG_TAILPOLL=0 timeout 200 python scratch/asg_phases.py 2>&1 | grep -v amdgpu | grep "checksum\|mean solve\|CERT\|SAP1_DONE"
for v in asg_nobuildvec asg_noprefetch; do echo "== $v"; CFM_LIB_OVERRIDE=scratch/variants/$v.so timeout 200 python scratch/asg_phases.py 2>&1 | grep -v amdgpu | grep "checksum\|mean solve\|SAP(ms\|BUILD"; done
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "assign or exact or golden or reference_suite or prefetch" 2>&1 | tail -3
