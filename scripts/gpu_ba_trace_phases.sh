#!/bin/bash
# rocprofv3 kernel trace of a cfg4 solve that first accepts its steps (iterations 1-41) and then, converged, rejects them:
# per kernel the mean duration and the mean gap to the kernel before it, for both phases (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ba_trace
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python3 $R/scripts/gpu_ba_bench_region.py > $O/log.txt 2>&1
F=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "") for r in rows]
st = [int(r["Start_Timestamp"]) for r in rows]; en = [int(r["End_Timestamp"]) for r in rows]
# iterations: from one ba_eliminate_mfma to the next
idx = [i for i, n in enumerate(names) if n.startswith("ba_eliminate_mfma")]
print("linearisations:", len(idx))
def phase(lo, hi, label):
    dur = collections.defaultdict(list); gap = collections.defaultdict(list); tot = []
    for a, b in zip(idx[lo:hi], idx[lo + 1:hi + 1]):
        tot.append(st[b] - st[a])
        for i in range(a, b):
            dur[names[i]].append(en[i] - st[i])
            gap[names[i]].append(st[i] - en[i - 1])
    print(f"{label}: {len(tot)} iterations, mean {sum(tot) / len(tot) / 1e3:.1f} us from one elimination's start to the next")
    for n in dur:
        print(f"   {n[:28]:28s} x{len(dur[n]) / len(tot):4.1f}  {sum(dur[n]) / len(dur[n]) / 1e3:7.1f} us   gap before {sum(gap[n]) / len(gap[n]) / 1e3:5.1f} us")
phase(4, 20, "early (iterations 5-20)")
phase(24, 40, "moving solve (iterations 25-40)")
phase(120, 200, "past convergence, radius 0 (iterations 121-200)")
PY
rm -rf $O/kt
