#!/bin/bash
# Round 6, GPU run 2: list-solver radius policy / eps0 sweep; kernel + HIP API trace of the slow regions (tools/tail_diag.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
BENCH_POOL=1 NINST=16 python tools/asg_sched_sweep.py "radius=0" "radius=30" "radius=50" "radius=75" "radius=100" "eps0=1.6e-2" "eps0=3.2e-2" "eps0=1.6e-2,radius=50" "theta=1.7,radius=50" "radius=0" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_sweep2.txt
O=gpurun_out/r6_tail; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/pub -- python bench.py --steps 20 --warmup 5 --public-only --no-cpu-baseline --repeats 21 > $O/pub.json.log 2>&1
python tools/tail_diag.py $O/pub > gpurun_out/r6_tail_public.txt 2>&1
rm -rf $O/pub
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/head -- python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --repeats 21 > $O/head.json.log 2>&1
python tools/tail_diag.py $O/head > gpurun_out/r6_tail_headline.txt 2>&1
rm -rf $O/head
tail -1 $O/pub.json.log | cut -c1-1500
