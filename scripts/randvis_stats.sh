# rocprofv3 kernel statistics of LM iterations at 200 cameras / 20 000 points with random visibility
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rv && mkdir -p /tmp/rv
rocprofv3 --kernel-trace --stats -d /tmp/rv -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_rand_vis_rate.py > /tmp/rv/log.txt 2>&1
grep "visibility" /tmp/rv/log.txt | cut -c1-60
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/rv/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-60s calls %6s  total %10.1f us  avg %8.1f us  %5s%%" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
