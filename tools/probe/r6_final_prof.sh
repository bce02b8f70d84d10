#!/bin/bash
# the round's evidence: profiles, then the two bench lines (the bench reads the newest PMC summaries from profiles/)
bash tools/profile_round.sh > gpurun_out/profile_round.log 2>&1
for f in gpurun_out/prof/*; do b=$(basename $f); cp $f profiles/r6_$b; done
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_steps20.json 2> gpurun_out/bench_steps20.err
tail -c 600 gpurun_out/bench_default.json; echo; tail -3 gpurun_out/bench_default.err
