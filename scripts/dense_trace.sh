# per-launch durations of chol_step2 over one LM run with the dense factorisation (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/dt && mkdir -p /tmp/dt
SFMHIP_BA_ND=0 rocprofv3 --kernel-trace -d /tmp/dt -o tr --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/gpu_dense_sizes.py ${1:-640} > /tmp/dt/log.txt 2>&1
grep -v rocprofv3 /tmp/dt/log.txt | tail -5; ls -R /tmp/dt | head -20
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/dt/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ch = [r for r in rows if 'chol_step2' in r['Kernel_Name']]
# the last full factorisation: find the run of launches with decreasing grid
seq = [(int(r['Grid_Size_X'] if 'Grid_Size_X' in r else r['Grid_Size']) // int(r['Workgroup_Size_X'] if 'Workgroup_Size_X' in r else r['Workgroup_Size']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in ch]
n = len(seq)
# launches per factorisation: count until the grid jumps up again
per = next((i for i in range(2, n) if seq[i][0] > seq[i - 1][0] + 50), n)
last = seq[-per:]
print("launches per factorisation", per, " sum of durations %.1f us" % (sum(s[1] for s in last) / 1e3), " span %.1f us" % ((last[-1][3] - last[0][2]) / 1e3))
for i, s in enumerate(last):
    if i % 4 == 0 or i == per - 1:
        gap = (last[i][2] - last[i - 1][3]) / 1e3 if i else 0
        print("k2 %3d  wgs %5d  %.1f us  (gap before %.1f)" % (i, s[0], s[1] / 1e3, gap))
PY
