#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 1200 python -m pytest tests/test_gpu_transport.py -x -q -m gpu -s > gpurun_out/r6_t5.log 2>&1
echo "transport tests rc=$?" >> gpurun_out/r6_t5.log
timeout 600 python tools/transport_bench.py > gpurun_out/r6_transport_bench.txt 2>&1
tail -30 gpurun_out/r6_t5.log; cat gpurun_out/r6_transport_bench.txt
