cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps -o ps --output-format csv -- python3 $R/scripts/gpu_score_time.py > /tmp/ps.log 2>&1
tail -5 /tmp/ps.log
F=$(find /tmp/ps -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:60].ljust(60), r["Calls"], r["TotalDurationNs"], r["AverageNs"])
PY
