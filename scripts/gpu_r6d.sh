#!/bin/bash
# round 6, job d: arena + pool + plan cache; whole GPU suite; bench line; adjustBundle call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
python scripts/gpu_adjust_bundle_call.py 4 > $O/adjust_bundle_call.txt 2>&1; cat $O/adjust_bundle_call.txt | cut -c1-330
SFMHIP_PROFILE_CREATE=1 python scripts/gpu_ba_create_time.py > $O/create.log 2>&1; tail -13 $O/create.log
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
python bench.py > $O/bench_full.json 2> $O/bench_full.err; tail -c 600 $O/bench_full.err
python - <<PY
import json
d=json.load(open("$O/bench_full.json"))
print({k:d[k] for k in ["value","ba_iterations_per_s","ms_per_step","ms_match_sweep","ms_ba_iteration"]})
print(d["roofline"]["frac"], d["roofline"]["launch_ms"], d["roofline_k2"], d["ba_amdahl"]["shard_measured"], d["ba_amdahl"]["bound_8gpu_speedup_measured"])
print(json.dumps(d["adjust_bundle_call"])[:1500]); print(d["config"]["ba_timed_iterations"], d["config"]["ba_steps_accepted_in_timed_iterations"])
PY
