#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 150 python scratch/asg_phases.py 2>&1 | grep -v amdgpu | grep "mean solve\|CERT\|checksum"
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "assign or exact or golden or reference_suite or wasserstein" 2>&1 | tail -2
