#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests/test_gpu_transport.py tests/test_gpu_kernels.py tests/test_gpu_golden.py -x -q -m gpu -s > gpurun_out/r6_t7.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_t7.log
timeout 900 python tools/transport_bench.py > gpurun_out/r6_transport_bench.txt 2>&1
tail -12 gpurun_out/r6_t7.log; cat gpurun_out/r6_transport_bench.txt
