# rocprofv3 kernel statistics of an LM run on a ring capture (argv: camera count, SFMHIP_BA_ND mode)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rs && mkdir -p /tmp/rs
cat > /tmp/rs/run.py <<PY
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sfm_danpipeline_amd import synth, bundle, _lib
ctx = _lib.default_context()
nc = int(sys.argv[1]); npt = {200: 100000, 400: 20000, 640: 12000, 1400: 9000}.get(nc, 10000)
pb = synth.ba_problem(nc, npt, 10 if nc == 200 else 8, seed=5)
prob = bundle.BaProblem(nc, npt, pb["obs_cam"], pb["obs_pt"], pb["obs_xy"], ctx=ctx)
prob.set_params(pb["cams0"], pb["pts0"], pb["focal0"])
prob.iterate(3)
t0 = time.time(); s = prob.iterate(20); dt = time.time() - t0
print(f"{nc} cameras / {npt} points: {20/dt:.1f} it/s  tree {prob.reduced_tree()}", flush=True)
PY
SFMHIP_BA_ND=${2:-2} rocprofv3 --kernel-trace --stats -d /tmp/rs -o st --output-format csv -- python3 /tmp/rs/run.py ${1:-640} > /tmp/rs/log.txt 2>&1
grep "it/s" /tmp/rs/log.txt
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/rs/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print("%-60s calls %6s  total %10.1f us  avg %8.1f us  %5s%%" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3, r['Percentage']))
PY
