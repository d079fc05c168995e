#!/bin/bash
# round 6, job h: the whole GPU suite, the profile of `bench.py --lean --cfg5-sample 2000`, the full bench line
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r6k
python -m pytest tests -m gpu -x -q -s > gpurun_out/r6k/gpu_tests.log 2>&1; tail -4 gpurun_out/r6k/gpu_tests.log; grep "spin time-outs" gpurun_out/r6k/gpu_tests.log
bash scripts/gpu_profile_bench.sh r6_v3 > gpurun_out/r6k/profile.log 2>&1; tail -5 gpurun_out/r6k/profile.log
python bench.py > gpurun_out/r6k/bench_full.json 2> gpurun_out/r6k/bench_full.err; tail -c 300 gpurun_out/r6k/bench_full.err
