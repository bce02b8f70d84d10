#!/bin/bash
# Round-5 sweep of the epsilon schedule (theta, eps0, eps_last, stop_frac) of the asynchronous auction over 40 instances
# (run through gpurun).  Output kept in profiles/r5_async_sweep.txt.
cd /root/repo
rm -f gpurun_out/r5_eps_sweep2.txt
for sc in "3,8e-3,1e-6,0.02" "2,8e-3,1e-6,0.02" "2.5,8e-3,1e-6,0.02" "3,8e-3,1e-7,0.02" "2.5,8e-3,1e-7,0.02" "3,8e-3,1e-6,0.01" "3,8e-3,3e-7,0.02" "3,1.5e-2,1e-6,0.02" "3,4e-3,1e-6,0.02" "3.5,8e-3,1e-6,0.02" "3,8e-3,1e-6,0.02"; do
  echo "== SCHED $sc" >> gpurun_out/r5_eps_sweep2.txt
  SCHED=$sc NINST=16 BENCH_POOL=1 ASYNC_LIST="2,16,4" timeout 120 python tools/asg_async_ab.py 1 4 2>&1 | grep -v amdgpu.ids | grep -E "lone solve|us per mode|batch of" >> gpurun_out/r5_eps_sweep2.txt
done
cat gpurun_out/r5_eps_sweep2.txt
