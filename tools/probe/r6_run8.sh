#!/bin/bash
# Round 6, GPU run 8: transport tests, full bench, tail diagnosis of the headline loop (kernel + HIP API trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/test_gpu_transport.py -x -q -m gpu > gpurun_out/r6_t8.log 2>&1
echo "transport tests rc=$?" >> gpurun_out/r6_t8.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r6_bench8.json 2> gpurun_out/r6_bench8.err
O=gpurun_out/r6_tail; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --hip-trace --output-format csv -d $O/head -- python bench.py --steps 20 --warmup 5 --no-legs --no-cpu-baseline --repeats 41 > $O/head.json.log 2>&1
python tools/tail_diag.py $O/head > gpurun_out/r6_tail_headline2.txt 2>&1
rm -rf $O/head
tail -3 gpurun_out/r6_t8.log
