#!/bin/bash
# rocprofv3 kernel statistics of the BA iteration at cfg4 (run through gpurun): scripts/gpu_prof_ba.sh <outdir>
out=${1:-gpurun_out/r2/prof_ba}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_ba -o ba --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_ba_iter_time.py cfg4 > $GRAFT_REPO_ROOT/$out/run.log 2>&1
f=$(find /tmp/prof_ba -name "*kernel_stats.csv" | head -1)
cp "$f" $GRAFT_REPO_ROOT/$out/ba_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
tail -2 $GRAFT_REPO_ROOT/$out/run.log
