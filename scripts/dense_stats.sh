# rocprofv3 kernel statistics of an LM run with the dense factorisation (argv: camera count)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ds && mkdir -p /tmp/ds
SFMHIP_BA_ND=0 rocprofv3 --kernel-trace --stats -d /tmp/ds -o st --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_dense_sizes.py ${1:-640} > /tmp/ds/log.txt 2>&1
grep "it/s" /tmp/ds/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ds/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-60s calls %6s  total %10.1f us  avg %8.1f us  %5s%%" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
