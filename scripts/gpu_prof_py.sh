#!/bin/bash
# rocprofv3 per-kernel averages of any python script of this repo (run through gpurun): gpu_prof_py.sh <script.py> [args...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_py
mkdir -p $O
s=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/$s "$@" > $O/log.txt 2>&1
F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:56].ljust(56), r["Calls"].rjust(6), f'{float(r["AverageNs"]) / 1e3:9.1f} us', f'{100 * float(r["TotalDurationNs"]) / tot:5.1f} %')
PY
tail -3 $O/log.txt
rm -rf $O/kt
