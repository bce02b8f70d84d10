#!/bin/bash
# Round 6 A/B: the list solver with / without the seeds of the surviving trees (interleaved, 40 instances x 2 each)
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in noseed seed; do
  echo "== $v"; NINST=16 BENCH_POOL=1 CFM_LIB_PATH=tools/probe/libcfm_$v.so python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -E "lone mean|identical"
done; done
