#!/bin/bash
# rocprofv3 per-kernel averages of a cfg4 LM iteration with the atomic scatter (mode 0, the default) and with the slab
# epilogue + ba_gather_slabs (mode 1: SFMHIP_BA_DETERMINISTIC=1); run through gpurun
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 0 1; do
  O=$R/gpurun_out/elim_mode$mode
  mkdir -p $O
  SFMHIP_BA_DETERMINISTIC=$mode timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/scripts/gpu_ba_iter_time.py cfg4 > $O/log.txt 2>&1
  F=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  echo "mode $mode"
  python3 - "$F" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:50].ljust(50), r["Calls"], r["AverageNs"])
PY
  rm -rf $O/kt
done
