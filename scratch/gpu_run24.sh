#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for b in 128 64; do for p in 1 3 4 6; do CFM_ASG_BLOCKS=$b timeout 200 python bench.py --pipeline $p --steps 40 --no-cpu-baseline --no-sinkhorn 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('blocks $b', d[\"config\"][\"schedule\"][:34], round(d[\"value\"]), round(d[\"ms_per_step\"],3), round(d[\"assign_ms_per_step\"],3), d[\"ms_per_step_sequential\"])"; done; done
