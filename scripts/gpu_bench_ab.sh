#!/bin/bash
# Same-box A/B of the bench line's matching half: `bench.py` (matching + BA timed regions, sustained, host-visible; no CPU / cfg5 /
# scoring / batch / whole-call legs) for the product and for other builds of the library, alternating.
# usage: gpu_bench_ab.sh <rounds> <so-or-"product"> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rounds=$1; shift
for i in $(seq $rounds); do
  for so in "$@"; do
    if [ "$so" = product ]; then p=""; else p=$R/sfm_danpipeline_amd/$so; fi
    SFMHIP_SO=$p python3 $R/bench.py --no-cpu-baseline --no-cfg5 --no-score --no-ba-batch --no-adjust-bundle 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-28s value %.0f  ms_match_sweep %.4f  launch_ms %.4f  sustained %.0f  host_visible %.0f  sweep_clock %.3f GHz  bare-MFMA %.0f TOPS  BA %.0f it/s' % ('$so', d['value'], d['ms_match_sweep'], r['launch_ms'], d['sustained']['pairs_per_s'], d['value_host_visible']['pairs_per_s'], r['sweep_clock_ghz'], r['sustained_peak'], d['ba_iterations_per_s']))"
  done
done
