#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { echo "--- handoff=$1 arr=$2 stop=$3 theta=$4"; timeout 100 python scratch/asg_pool.py 8 $1 $2 $3 $4 2>&1 | grep -v amdgpu | awk '{s+=$3; n++; printf "%s ", $3} END {printf " | mean %.2f ms\n", s/n}'; }
run 6 15 0.02 5
run 6 15 0.03 5
run 6 15 0.04 5
run 6 10 0.03 5
run 6 20 0.03 5
run 8 15 0.03 5
run 4 15 0.03 5
