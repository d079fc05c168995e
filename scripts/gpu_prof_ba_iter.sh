#!/bin/bash
# rocprofv3 kernel statistics of scripts/gpu_ba_iter_time.py cfg4 (run through gpurun); env passes through (e.g. SFMHIP_BA_BACKSUB_WPP)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bi
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bi -o bi --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_ba_iter_time.py cfg4 > /tmp/bi.log 2>&1
f=$(find /tmp/prof_bi -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
