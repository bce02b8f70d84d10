#!/bin/bash
# Round 6 A/B of list-solver variants (interleaved, 40 instances x 2 each): CFM_AB_VARIANTS="seed carry4 ..." names
# tools/probe/libcfm_<v>.so (tools/probe/build_variant.sh); without it the product library is measured twice.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in ${CFM_AB_VARIANTS:-product}; do
  echo "== $v"
  if [ "$v" = product ]; then NINST=16 BENCH_POOL=1 python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -E "lone mean|identical"
  else NINST=16 BENCH_POOL=1 CFM_LIB_PATH=tools/probe/libcfm_$v.so python tools/asg_sched_sweep.py "theta=2.5" 2>&1 | grep -E "lone mean|identical"; fi
done; done
