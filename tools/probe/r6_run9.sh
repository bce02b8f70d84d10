#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/probe/blocking_sync_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r6_blocking_probe.txt
for S in 0 4 8 32; do CFM_WGRAD_S=$S python tools/gemm_quick.py 2>&1 | grep -v amdgpu | sed "s/^/S=$S /" >> gpurun_out/r6_wgrad_split.txt; done
for S in 0 8; do CFM_WGRAD_S=$S python tools/gemm_quick.py 2>&1 | grep -v amdgpu | sed "s/^/S=$S /" >> gpurun_out/r6_wgrad_split.txt; done
cat gpurun_out/r6_blocking_probe.txt gpurun_out/r6_wgrad_split.txt
