#!/bin/bash
# First GPU call of the next round (≈ 2 min on the box): the cost matrix on the second-generation direct-to-LDS loop
# (csrc/gemm_glds.h: gl_run_padded; csrc/cost.hip: COST_GLDS_V2) against the default engine.
#   1. build the variant libraries HERE (no GPU needed):   bash tools/probe/try_glds_v2.sh build
#   2. on the GPU box (gpurun):                             bash tools/probe/try_glds_v2.sh run
# `run` checks bit-equality of the cost matrices (tests/test_gpu_glds.py against the variant library) and times cost /
# forward / regression step at the C3 shapes (tools/gemm_quick.py) for: the product library, the variant with and without
# the alternating issue priority.  Adoption = make COST_GLDS_V2 and CFM_COST_GLDS the defaults, then the full GPU suite.
set -e
R="$(cd "$(dirname "$0")/../.." && pwd)"
case "$1" in
  build)
    bash "$R/tools/probe/build_variant_all.sh" gv2 "-DCOST_GLDS_V2=1" cost
    bash "$R/tools/probe/build_variant_all.sh" gv2nofair "-DCOST_GLDS_V2=1 -DCOST_GLDS_FAIR=0" cost
    bash "$R/tools/probe/build_variant_all.sh" gv2plain "-DCOST_GLDS_V2=1 -DCOST_GLDS_FAIR=0 -DCOST_GLDS_LOOP=0" cost
    bash "$R/tools/probe/build_variant_all.sh" gcfair "-DGC_FAIR=1" cost mlp mlp_train      # the register-staged core with alternating priority
    bash "$R/tools/probe/build_variant_all.sh" gcpipe "-DGC_PIPE=1" cost mlp mlp_train ode   # ... with the pipelined K-step boundary (same bits by construction)
    bash "$R/tools/probe/build_variant_all.sh" gcpipefair "-DGC_PIPE=1 -DGC_FAIR=1" cost mlp mlp_train ode
    # the bid rounds with the state block, the control record and the bidder counters in the FIRST batch of loads (assign.hip:
    # ASG_PREFETCH_CTL; the default code spends ~6 dependent round trips on them per round where one would do)
    bash "$R/tools/probe/build_variant_all.sh" asgpf "-DASG_PREFETCH_CTL=1" assign
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off "$R/tools/probe/glds_probe.hip" -o "$R/tools/probe/glds_probe"
    ;;
  run)
    cd "$R"; mkdir -p gpurun_out
    {
      timeout 60 tools/probe/glds_probe padded
      timeout 60 tools/probe/glds_probe ab 9          # interleaved A/B of every loop, with each launch's clock: which gains are real?
      timeout 60 tools/probe/glds_probe ab 9 4096 4096 800
      timeout 60 tools/probe/glds_probe               # timelines and wait shares of the instrumented variants
      for V in gv2 gv2nofair gv2plain; do
        echo "== bit-equality with the default engine, variant $V"
        CFM_LIB_PATH=tools/probe/libcfm_$V.so timeout 300 python -m pytest tests/test_gpu_glds.py -q -x -p no:cacheprovider 2>&1 | tail -3
      done
      echo "== timings (C3 shapes)"
      timeout 120 python tools/gemm_quick.py
      for V in gv2 gv2nofair gv2plain; do CFM_COST_GLDS=1 CFM_LIB_PATH=tools/probe/libcfm_$V.so timeout 120 python tools/gemm_quick.py; done
      for V in gcfair gcpipe gcpipefair; do CFM_LIB_PATH=tools/probe/libcfm_$V.so timeout 120 python tools/gemm_quick.py; done
      # the layers on 128 x 64 tiles (one workgroup per CU: the lone-wave regime the pipelined boundary helps most)
      CFM_GEMM_TILE=1 timeout 120 python tools/gemm_quick.py; CFM_GEMM_TILE=1 CFM_LIB_PATH=tools/probe/libcfm_gcpipe.so timeout 120 python tools/gemm_quick.py
      echo "== the dense-product tests on the pipelined register-staged core (bit-exactness against the references they hold)"
      CFM_LIB_PATH=tools/probe/libcfm_gcpipe.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_glds.py tests/test_gpu_fullsize.py tests/test_gpu_golden.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
    } 2>&1 | tee gpurun_out/try_glds_v2.txt
    ;;
  run2)
    # second call: the exact solver with ASG_PREFETCH_CTL — its tests, then solve_ms / avg_launch_us / ms_per_step next to the product's
    cd "$R"; mkdir -p gpurun_out
    {
      CFM_LIB_PATH=tools/probe/libcfm_asgpf.so timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_assign_batch.py tests/test_gpu_fullsize.py tests/test_gpu_round2.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -3
      for L in "" tools/probe/libcfm_asgpf.so; do
        CFM_LIB_PATH=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('$L' or 'product', 'ms_per_step', round(d['ms_per_step'],4), 'seq', round(d['ms_per_step_sequential'],3), 'solve_ms', round(r['solve_ms'],3), 'avg_launch_us', round(r['avg_launch_us'],2), 'steady', round(d['steady_state']['ms_per_step'],4))"
      done
    } 2>&1 | tee gpurun_out/try_asgpf.txt
    ;;
  *) echo "usage: $0 build | run | run2"; exit 2;;
esac
