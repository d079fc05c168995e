#!/bin/bash
# rocprofv3 kernel statistics of the BA linearisation with random visibility (run through gpurun)
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_rv -o rv --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rand_vis.py > /tmp/rv.log 2>&1
f=$(find /tmp/prof_rv -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
